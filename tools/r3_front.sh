set -e
mkdir -p gpurun_out/r3b
PP_PLAN_TUNE=front_pad_frac=2.0 timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3b/pytest2.log 2>&1 || { tail -40 gpurun_out/r3b/pytest2.log; exit 1; }
tail -3 gpurun_out/r3b/pytest2.log
PP_PLAN_TUNE=front_pad_frac=2.0,front_scale_rows=8 PP_LIB_VARIANT=stamps python tools/stamp_front.py 2>&1 | grep -v amdgpu.ids
bash tools/r3_tune.sh base:front_max=4 base:front_pad_frac=2.0,front_scale_rows=8 base:front_pad_frac=2.0,front_scale_rows=4 base:front_pad_frac=2.0,front_scale_rows=16
