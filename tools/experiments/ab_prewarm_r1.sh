# first proof of a fresh process behind sp_prewarm whose full-size round 1 covers a fraction of the columns (SP_PREWARM_R1_FRAC)
for i in 1 2 3; do for f in 1.0 0.5 0.2 0.05; do for shape in "149000 8" "70000 4"; do
SP_PREWARM_R1_FRAC=$f python bench.py --cold-child /tmp/c.json --cold-shape $shape --cold-path run+prewarm 2>/dev/null >/dev/null
echo "frac $f $shape: $(python -c "import json;d=json.load(open('/tmp/c.json'));print({k:round(d[k],1) for k in ('first_call_ms','second_call_ms','third_call_ms','prewarm_ms','front_end_and_prewarm_ms','front_end_run_ms')})")"
done; done; done
