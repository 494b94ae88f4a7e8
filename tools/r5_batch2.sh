cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_population.py tests/test_gpu_kernels.py -x -q -m gpu -k "separable or sepf or frame or two_pass or forced" 2>&1 | tail -15 > gpurun_out/r5_t6.log
python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "stress" 2>&1 | tail -15 >> gpurun_out/r5_t6.log
rocprofv3 --kernel-trace --stats -d gpurun_out/r5_c5s_prof -o c -- python3 tools/cfg_loop.py C5S 12 > gpurun_out/r5_c5s.log 2>&1
python - <<'PY' > gpurun_out/r5_c5s_kern.txt
import sqlite3
db=sqlite3.connect('gpurun_out/r5_c5s_prof/c_results.db')
rows=list(db.execute("select name, count(*), avg(end-start)/1e3, min(end-start)/1e3 from kernels group by name order by 3 desc limit 14"))
for r in rows: print("%-70s n=%4d avg=%8.1f us min=%8.1f"%(r[0][:70],r[1],r[2],r[3]))
PY
python bench.py --no-cpu-baseline --no-map --no-mcmc --no-stim-map --steps 5 > gpurun_out/r5_bench_b.json 2> gpurun_out/r5_bench_b.err
cat gpurun_out/r5_t6.log; tail -2 gpurun_out/r5_c5s.log; cat gpurun_out/r5_c5s_kern.txt; python -c "
import json; d=json.load(open('gpurun_out/r5_bench_b.json')); print(json.dumps(d.get('secondary_stim'), indent=1)[:1500])"
