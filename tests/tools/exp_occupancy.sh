set -e
cd ffpic_amd/csrc
for cfg in "" "-DWAVES_PER_WG=2" "-DWAVES_PER_WG=8"; do
  rm -f ffhip_jpeg.o; make -s HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function $cfg" >/dev/null 2>&1
  echo "== cfg [$cfg]"; (cd ../.. && for v in 10 12 13; do FFHIP_JPEG_VARIANT=$v python tests/tools/time_kernel.py --steps 10 --rounds 3 2>/dev/null | grep variant; done)
done
rm -f ffhip_jpeg.o; make -s >/dev/null 2>&1
