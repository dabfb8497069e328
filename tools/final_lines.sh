#!/bin/bash
# The round's closing measurement on ONE box, after tools/profile_all.sh <tag> and with its summaries committed under profiles/:
# the bench lines, the BER sweeps and the stream figures.   bash tools/final_lines.sh r05
set -e
tag=$1
bash tools/bench_lines.sh $tag
out=gpurun_out/${tag}_lines
python3 tools/ber_sweep.py --json-out $out/${tag}_ber_sweep.json > /dev/null
python3 tools/ber_sweep.py --detector PAM --json-out $out/${tag}_ber_sweep_pam.json > /dev/null
python3 tools/ber_sweep.py --waveform multih --json-out $out/${tag}_ber_sweep_multih.json > /dev/null
python3 tools/ber_sweep.py --waveform multih --states 64 --json-out $out/${tag}_ber_sweep_multih64.json > /dev/null
python3 tools/ber_sweep.py --waveform multih --states 256 --passes 1 --json-out $out/${tag}_ber_sweep_multih256.json > /dev/null
python3 tools/ber_sweep.py --waveform pcmfm --json-out $out/${tag}_ber_sweep_pcmfm.json > /dev/null
echo SWEEPSDONE
python3 tools/stream_bench.py --pipelined --chunk 8388608 > $out/${tag}_stream_soqpsk.json
# BASELINE configs[4] names hipGraph replay: the stream's steady-state chunk replayed as ONE graph, and the two-stream chunk pipeline
# captured as graphs of 2 / 4 / 8 interior chunks — beside the eager forms (round-5 verdict, item 2)
python3 tools/stream_bench.py --chunk 8388608 > $out/${tag}_stream_soqpsk_eager.json
python3 tools/stream_bench.py --graph --chunk 8388608 > $out/${tag}_stream_soqpsk_graph.json
python3 tools/stream_bench.py --graph-pipelined 2 --chunk 8388608 > $out/${tag}_stream_soqpsk_graph_pipe2.json
python3 tools/stream_bench.py --graph-pipelined 4 --chunk 8388608 > $out/${tag}_stream_soqpsk_graph_pipe4.json
python3 tools/stream_bench.py --graph-pipelined 8 --chunk 8388608 > $out/${tag}_stream_soqpsk_graph_pipe8.json
python3 tools/stream_bench.py --pipelined --chunk 8388608 --detector PAM > $out/${tag}_stream_pam.json
python3 tools/stream_bench.py --pipelined --chunk 10485760 --waveform multih > $out/${tag}_stream_multih.json
python3 tools/stream_bench.py --pipelined --chunk 10485760 --waveform pcmfm > $out/${tag}_stream_pcmfm.json
python3 tools/stream_bench.py --chunk 4194304 --waveform multih > $out/${tag}_stream_multih_eager.json
python3 tools/stream_bench.py --chunk 4194304 --waveform pcmfm > $out/${tag}_stream_pcmfm_eager.json
echo STREAMSDONE
