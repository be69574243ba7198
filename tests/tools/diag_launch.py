import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from ffpic_amd import capi, ops, shard, synth
import bench
dev = torch.device("cuda", 0)
L = capi.require_device(0)
for wl in ("c3", "c2"):
    cols, rows, n, _ = bench.WORKLOADS[wl]
    geom = capi.jpeg_geom(cols, rows)
    H, W = geom.height, geom.width
    t_y, t_u, t_v, q = bench.gen_batch_on_device(dev, cols, rows, n, 0)
    t_q = torch.from_numpy(q.astype(np.int16)).to(dev)
    out = torch.empty(n * W * 4 * H, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def step():
        ops.jpeg_recon_batch(geom, n, t_y.data_ptr(), t_u.data_ptr(), t_v.data_ptr(), t_q.data_ptr(), 0, out.data_ptr(), W*4, W*4*H, None, 0, st)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    r = shard.gather_status(0, n, 0, device=dev)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(wl, "launch loop %.2f ms, to sync %.2f ms (%.3f ms/step), gather %.2f ms" % ((t1-t0)*1e3, (t2-t0)*1e3, (t2-t0)*1e3/20, (t3-t2)*1e3))
    # one at a time
    ts=[]
    for _ in range(5):
        a=time.perf_counter(); step(); torch.cuda.synchronize(); ts.append((time.perf_counter()-a)*1e3)
    print("  single step+sync ms:", ["%.3f"%x for x in ts])
    del t_y, t_u, t_v, out
