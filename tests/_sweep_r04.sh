mkdir -p gpurun_out/r04/sw; rm -f gpurun_out/r04/sw/res.txt
run() { name=$1; shift; env "$@" python bench.py --secondary 0 --cpu-seconds 0.3 --extras 0 --steps 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['config']['launches_per_apply'])" >> gpurun_out/r04/sw/res.txt; }
run base X=1
run cs_sparse HIFIR_AMD_CS_SPARSE=1
run base2 X=1
cat gpurun_out/r04/sw/res.txt
