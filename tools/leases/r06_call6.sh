mkdir -p gpurun_out
for t in "tests/test_graph_step.py::test_failed_capture_falls_back_to_eager_and_the_run_goes_on" "tests/test_graph_step.py::test_a_failing_capture_leaves_the_stepper_usable" "tests/test_graph_step.py::test_rccl_collectives_inside_the_captured_step" "tests/test_step_glue.py::test_bench_launches_its_own_ranks" "tests/test_bench_dry_run.py::test_dry_run_bucket_is_the_real_flat_bucket"; do
  n=$(echo $t | sed 's/.*:://' | cut -c1-40)
  timeout 400 python -X faulthandler -m pytest "$t" -m gpu -x -q -s > gpurun_out/r06_t_$n.txt 2>&1
  echo "== $t rc=$?"; grep -E "passed|failed|Fatal|Error|error:" gpurun_out/r06_t_$n.txt | head -5 | cut -c1-300
done
timeout 300 python -m pytest tests/test_hip_edge_cases.py tests/test_hip_fullsize_train.py -m gpu -q -k "one_launch or small_batch or chunked_oracle" 2>&1 | tail -15 | cut -c1-400
