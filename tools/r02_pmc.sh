O=gpurun_out/r02g; mkdir -p $O; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python bench.py --no-cpu --no-stagei --no-strong --no-sequential --steps 1 --warmup 1 --seeds 1000 > /dev/null 2> $O/pmc_${c}_err.txt
done
python - <<'PY'
import csv,glob
for c in ('FETCH_SIZE','WRITE_SIZE'):
    fn=glob.glob(f'gpurun_out/r02g/pmc_{c}/**/*counter_collection.csv', recursive=True)[0]
    rows=[r for r in csv.DictReader(open(fn)) if 'k_chain_solve' in r['Kernel_Name']]
    print(c, [(int(r['Grid_Size'])//256, round(float(r['Counter_Value'])/1e3,1)) for r in rows])
PY
