python -m pytest tests/test_cpm_detector.py tests/test_lifecycle.py -m gpu -q -x > gpurun_out/r04_f_pytest.log 2>&1; echo "pytest(default) rc=$?"; tail -3 gpurun_out/r04_f_pytest.log
WF_CPM_LANE_R=2 python -m pytest tests/test_cpm_detector.py -m gpu -q -x > gpurun_out/r04_f_pytest_r2.log 2>&1; echo "pytest(R2) rc=$?"; tail -3 gpurun_out/r04_f_pytest_r2.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for wf in multih pcmfm; do
 for cfg in "L0:WF_CPM_LANES=0" "R2:WF_CPM_LANE_R=2" "R3:WF_CPM_LANE_R=3" "R4:WF_CPM_LANE_R=4"; do
  tag=${cfg%%:*}; e1=${cfg#*:}
  env $e1 python bench.py --waveform $wf --fuse 47 --no-cpu-baseline --steady-steps 300 --overlap-streams 0 > gpurun_out/r04_f_bench_${wf}_${tag}_f47.json 2> gpurun_out/r04_f_bench_${wf}_${tag}_f47.err
  export $e1
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_f_kt_${wf}_${tag} -o kt -- python3 bench.py --waveform $wf --fuse 15 --no-cpu-baseline --steady-steps 0 --overlap-streams 0 --steps 5 > /dev/null 2> gpurun_out/r04_f_kt_${wf}_${tag}.err
  unset WF_CPM_LANES WF_CPM_LANE_R
  find gpurun_out/r04_f_kt_${wf}_${tag} -name '*kernel_trace.csv' -delete
 done
done
echo done
