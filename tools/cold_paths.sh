python -m pytest tests/test_gpu_prover.py -x -q -k "prewarm" 2>&1 | tail -5
for shape in "149000 8" "70000 4"; do for path in run rows run+prewarm rows+prewarm run+prewarm; do
python bench.py --cold-child /tmp/c.json --cold-shape $shape --cold-path $path 2>/dev/null >/dev/null; echo "$shape $path: $(cat /tmp/c.json)"; done; done
