mkdir -p gpurun_out/r04/sw; rm -f gpurun_out/r04/sw/res.txt
run() { name=$1; shift; env "$@" python bench.py --secondary 0 --cpu-seconds 0.3 --extras 0 --steps 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['config']['launches_per_apply'])" >> gpurun_out/r04/sw/res.txt; }
run base X=1
run a0late HIFIR_AMD_CD_DBG=256
run base2 X=1
run a0late2 HIFIR_AMD_CD_DBG=256
cat gpurun_out/r04/sw/res.txt
