WF_CPM_LANES=1 python -m pytest tests/test_cpm_detector.py -m gpu -q -x -k "not full_size" > gpurun_out/r04_n_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r04_n_pytest.log
export BENCH_FLAGS="--waveform multih --steps 10"
tools/ab_kernel.sh wf_cpm_lanes.hip "" "SRC=waveforms_amd/csrc/build/old_wf_cpm_lanes.hip" 2>&1 | grep viterbi
export BENCH_FLAGS="--waveform pcmfm --steps 10"
tools/ab_kernel.sh wf_cpm_lanes.hip "" "SRC=waveforms_amd/csrc/build/old_wf_cpm_lanes.hip" 2>&1 | grep viterbi
