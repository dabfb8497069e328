#!/bin/bash
# Profile `bench.py` on the GPU box: kernel trace + stats, then FETCH_SIZE, WRITE_SIZE and the SQ
# instruction counters in their own --pmc passes (never combined with other trace domains).
#   tools/profile.sh <tag> [bench.py flags...]      e.g. tools/profile.sh r01_fused --fuse 3
# Writes gpurun_out/<tag>/{kt,fetch,write}/... and profiles/<tag>_{kernel_stats.csv,summary.json,
# bench_under_rocprof.json} (copy profiles/ back from gpurun_out/<tag>/profiles on the host).
set -e
tag=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/$tag
mkdir -p $out/profiles
# --overlap-streams 0: only the single-stream timed region, so the per-kernel averages are those of kernels running alone
flags="--steps 5 --warmup 2 --no-cpu-baseline --overlap-streams 0 --steady-steps 0 --ber-points none $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py $flags > $out/profiles/${tag}_bench_under_rocprof.json 2> $out/kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o fetch -- python3 bench.py $flags > /dev/null 2> $out/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o write -- python3 bench.py $flags > /dev/null 2> $out/write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/sq -o sq -- python3 bench.py $flags > /dev/null 2> $out/sq.err
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/mfma -o mfma -- python3 bench.py $flags > /dev/null 2> $out/mfma.err
ks=$(find $out/kt -name '*kernel_stats.csv' | head -1)
mc=$(find $out/mfma -name '*counter_collection.csv' | head -1)
fc=$(find $out/fetch -name '*counter_collection.csv' | head -1)
wc=$(find $out/write -name '*counter_collection.csv' | head -1)
sc=$(find $out/sq -name '*counter_collection.csv' | head -1)
cp $ks $out/profiles/${tag}_kernel_stats.csv
kt=$(find $out/kt -name '*kernel_trace.csv' | head -1)
sps=8; case " $* " in *" --sps 10 "*) sps=10;; *" --sps 20 "*) sps=20;; esac
python3 tools/pmc_summary.py $tag $ks $fc $wc --sq $sc --mfma $mc --trace $kt --nsym 10000000 --sps $sps --out $out/profiles
# the bench line committed next to the summary is taken AFTER it, so that its roofline object quotes this
# very profile (traffic, VALU instructions, shader cycles) and says traffic_profile_matches_build: true
cp $out/profiles/${tag}_summary.json profiles/
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt2 -o kt2 -- python3 bench.py $flags > $out/profiles/${tag}_bench_under_rocprof.json 2> $out/kt2.err
# keep the merge-back small: the raw per-dispatch CSVs are tens of MB
find $out -name '*counter_collection.csv' -delete; find $out -name '*kernel_trace.csv' -delete
python3 - <<PY
import csv
for r in csv.DictReader(open("$out/profiles/${tag}_kernel_stats.csv")):
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(4), f'{float(r["AverageNs"])/1e3:9.2f} us', r["Percentage"])
PY
