# tools/switch_matrix.sh -- the parity and scale tests under the switches that select other code paths (GPU box).
# Expected: everything passes, except that without the second-level partition (DFK_SPLIT_FROM_LOG2=20) the forced
# ONE-pass run of test_hot_buckets_beyond_2_pow_31_instances_in_one_pass ends in DFK_E_NOMEM -- 3.6e9 instances in HBM
# tables are 186 GB -- loudly, as it should.  (No -x: the tests behind an expected failure run too.)
run() { echo "== $*" >> gpurun_out/matrix.log; env "$@" python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -q -m gpu 2>&1 | tail -4 >> gpurun_out/matrix.log; }
rm -f gpurun_out/matrix.log
run DFK_SPLIT_FROM_LOG2=20
run DFK_SPLIT_FROM_LOG2=20 DFK_MAX_SUBPASS_LOG2=0
run DFK_NO_OVERLAP=1
run DFK_DEFER_SWEEP_BELOW=0
run DFK_PLAN_GROWTH=1.3 DFK_PLAN_FIRST=16
