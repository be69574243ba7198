set -e
cd ffpic_amd/csrc
for cfg in "" "-DFFHIP_LDS_PAD=1024" "-DFFHIP_LDS_PAD=2816" "-DWAVES_PER_WG=2" "-DWAVES_PER_WG=8"; do
  rm -f ffhip_jpeg.o; make -s HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function $cfg" >/dev/null 2>&1
  echo "== cfg [$cfg]"; (cd ../.. && python tests/tools/time_kernel.py --steps 10 --rounds 3 2>/dev/null | grep variant)
done
rm -f ffhip_jpeg.o; make -s >/dev/null 2>&1
