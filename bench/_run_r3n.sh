set -o pipefail
D=gpurun_out/r3n; mkdir -p $D
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --batch 1024 --steps 30 --warmup 10 > $D/$name.json 2> $D/$name.err; echo "$name rc=$? $(grep -o '"ms_per_step": [0-9.]*' $D/$name.json)"; }
run base PDT_X=0
H=ffffffff
run low128 PDT_CU_MASK_WGRAD=$H,$H,$H,$H
run alt PDT_CU_MASK_WGRAD=55555555,55555555,55555555,55555555,55555555,55555555,55555555,55555555
run low192 PDT_CU_MASK_WGRAD=$H,$H,$H,$H,$H,$H
run q8 PDT_CU_MASK_WGRAD=0f0f0f0f,0f0f0f0f,0f0f0f0f,0f0f0f0f,0f0f0f0f,0f0f0f0f,0f0f0f0f,0f0f0f0f
run base2 PDT_X=0
