R=$GRAFT_REPO_ROOT
export CLIPENC_LIB_PATH=$R/clip_assisted_data_labeling_amd/libclipenc_hip_diag.so
for i in 1 2 3; do
  for v in unfused fused; do
    if [ $v = unfused ]; then export CLIPENC_FP8_UNFUSED=1; else unset CLIPENC_FP8_UNFUSED; fi
    timeout -k 10 200 python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --dtype fp8 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('$v', d['value'], {a.replace('gemm_fp8_kernel','f8').replace('attn_stream_kernel','attn'): round(b,1) for a,b in k.items() if b>0.2})"
  done
done
