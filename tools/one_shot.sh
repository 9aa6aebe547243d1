for i in 1 2 3; do for shape in "149000 8" "70000 4"; do for path in run run+prewarm run+ctx+prewarm; do
python bench.py --cold-child /tmp/c.json --cold-shape $shape --cold-path $path 2>/tmp/err.txt >/dev/null || tail -3 /tmp/err.txt
echo "$shape $path: $(python -c "import json;d=json.load(open('/tmp/c.json'));print({k:round(d[k],1) if isinstance(d[k],float) else d[k] for k in ('context_to_first_proof_ms','context_create_ms','front_end_run_ms','prewarm_ms','front_end_and_prewarm_ms','first_call_ms','third_call_ms')})")"
done; done; done
