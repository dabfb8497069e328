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
python3 bench.py --waveform pcmfm --no-cpu-baseline > $out/${tag}_bench_pcmfm.json
echo LINESDONE
