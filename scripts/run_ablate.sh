# developer ablation timings of the screened scan (needs a -DPG_SCAN_VARIANTS build); results are wrong by design
# VAR: 0 product, 1 no screen test, 2 no MFMA and no test (stream only), 4 test but never the hit path
# build: make -C pairec_amd/csrc -B CXXFLAGS='-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DPG_SCAN_VARIANTS' (and rebuild normally afterwards)
for b in ${BATCHES:-256}; do
for v in 0 4 1 2; do
  PG_SCREEN_VAR=$v PG_DEBUG_SCAN=1 python bench.py --steps 3 --warmup 1 --batch $b --no-cpu-baseline --latency-reqs 0 --no-extras --contexts 1 2>&1 | grep "plan 0 scan launch 2" | tail -2 | sed "s/^/batch $b var $v: /"
done; done
