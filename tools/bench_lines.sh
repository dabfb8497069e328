# The bench lines stored under profiles/ beside the rocprof summaries: default workload, PAM detector, sps 10 (both detectors) and
# the two CPM links, as the driver runs them (no flags but the workload's); then the BER sweep.
#   bash tools/bench_lines.sh <round tag, e.g. r03>     -> gpurun_out/<tag>_lines/*.json
set -e
tag=$1
out=gpurun_out/${tag}_lines
mkdir -p $out
python3 bench.py > $out/${tag}_bench.json
python3 bench.py --detector PAM --no-cpu-baseline > $out/${tag}_bench_pam.json
python3 bench.py --sps 10 --no-cpu-baseline > $out/${tag}_bench_sps10.json
python3 bench.py --sps 10 --detector PAM --no-cpu-baseline > $out/${tag}_bench_pam10.json
python3 bench.py --waveform multih --no-cpu-baseline > $out/${tag}_bench_multih.json
python3 bench.py --waveform multih --states 64 --no-cpu-baseline > $out/${tag}_bench_multih64.json
python3 bench.py --waveform multih --states 256 --steady-steps 100 --no-cpu-baseline > $out/${tag}_bench_multih256.json
python3 bench.py --fuse 0 --no-cpu-baseline > $out/${tag}_bench_unfused.json
python3 bench.py --waveform pcmfm --no-cpu-baseline > $out/${tag}_bench_pcmfm.json
# low Eb/N0, where chunk warm-ups do not merge and repairs cross chunk boundaries (round-4 verdict item 1): a line, not an exception
python3 bench.py --waveform pcmfm --ebn0 2 --steady-steps 200 --ber-points none --no-cpu-baseline > $out/${tag}_bench_pcmfm_2db.json
python3 bench.py --waveform pcmfm --ebn0 -8 --steady-steps 200 --ber-points none --no-cpu-baseline > $out/${tag}_bench_pcmfm_m8db.json
python3 bench.py --waveform multih --ebn0 0 --steady-steps 200 --ber-points none --no-cpu-baseline > $out/${tag}_bench_multih_0db.json
python3 bench.py --ebn0 -8 --steady-steps 200 --ber-points none --no-cpu-baseline > $out/${tag}_bench_m8db.json
echo LINESDONE
