# developer experiment (needs an experimental build that is not in the tree: persistent workgroups + atomic ticket in lm_fused_kernel,
# grid size from CPPF_QUEUE_WGS; kept for the record of profiles/r3_wave_timeline.txt)
set -e
for q in 0 1024 896 768 640 512; do
  echo -n "queue WGs $q:  "
  CPPF_QUEUE_WGS=$q CPPFLOW_HIP_LIB=build_var/lib_queue.so python bench.py --no-cpu-baseline --no-siblings --repeats 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('us/step %.2f   isolated kernel %.2f us  conv %.4f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['config']['converged_frac_pos_err_lt_1e-4']))"
done
echo -n "in-tree library: "; python bench.py --no-cpu-baseline --no-siblings --repeats 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('us/step %.2f   isolated kernel %.2f us  conv %.4f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['config']['converged_frac_pos_err_lt_1e-4']))"
