python -m pytest tests/test_cpm_detector.py -m gpu -q -x > gpurun_out/r04_m_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04_m_pytest.log
for wf in multih pcmfm; do
  python tools/stream_bench.py --waveform $wf --total 1e9 --chunk $((1<<22)) > gpurun_out/r04_m_stream_${wf}_eager.json 2> gpurun_out/r04_m_stream_${wf}_eager.err
  python tools/stream_bench.py --waveform $wf --total 1e9 --chunk $((1<<22)) --pipelined > gpurun_out/r04_m_stream_${wf}_piped.json 2> gpurun_out/r04_m_stream_${wf}_piped.err
  python tools/stream_bench.py --waveform $wf --total 1e9 --chunk $((1<<23)) --pipelined > gpurun_out/r04_m_stream_${wf}_piped23.json 2> gpurun_out/r04_m_stream_${wf}_piped23.err
  python tools/stream_bench.py --waveform $wf --total 1e9 --chunk 10485760 --pipelined > gpurun_out/r04_m_stream_${wf}_piped1e7.json 2> gpurun_out/r04_m_stream_${wf}_piped1e7.err
  tail -qn1 gpurun_out/r04_m_stream_${wf}_eager.json gpurun_out/r04_m_stream_${wf}_piped.json gpurun_out/r04_m_stream_${wf}_piped23.json gpurun_out/r04_m_stream_${wf}_piped1e7.json | cut -c1-330
done
