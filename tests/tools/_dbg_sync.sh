set -u
for b in 1024 2048 4096; do for parts in 8 4 2; do
  echo "bits $b parts $parts: $(FFHIP_JPEG_SYNC_PARTS=$parts FFHIP_JPEG_SYNC_BITS=$b python3 tests/tools/bench_huff_plain.py 2>&1 | tail -1)"
done; done
