# Every rocprofv3 profile the round commits (GPU box): tools/profile.sh per workload, blocks one after the other
# (--fuse 15: every kernel alone on the chip).   bash tools/profile_all.sh <round tag, e.g. r04>
set -e
tag=${1:-r05}
bash tools/profile.sh ${tag}_fuse15 --fuse 15 > gpurun_out/p_fuse15.log 2>&1
# every stage its own kernel (fir, phase, awgn, bank): the HBM-bound stage kernels of SURVEY 8(d) at the current build
bash tools/profile.sh ${tag}_unfused --fuse 0 > gpurun_out/p_unfused.log 2>&1
bash tools/profile.sh ${tag}_pam --detector PAM --fuse 15 > gpurun_out/p_pam.log 2>&1
bash tools/profile.sh ${tag}_sps10 --sps 10 --fuse 15 > gpurun_out/p_sps10.log 2>&1
bash tools/profile.sh ${tag}_pam10 --sps 10 --detector PAM --fuse 15 > gpurun_out/p_pam10.log 2>&1
bash tools/profile.sh ${tag}_multih --waveform multih --fuse 143 > gpurun_out/p_multih.log 2>&1
# the same link with its matched-filter rows in HBM (fuse bits 0-3 + the paired templates: the round-5 form), for the record
bash tools/profile.sh ${tag}_multih_rows --waveform multih --fuse 15 > gpurun_out/p_multih_rows.log 2>&1
bash tools/profile.sh ${tag}_multih64 --waveform multih --states 64 --fuse 15 > gpurun_out/p_multih64.log 2>&1
bash tools/profile.sh ${tag}_multih256 --waveform multih --states 256 --fuse 143 > gpurun_out/p_multih256.log 2>&1
bash tools/profile.sh ${tag}_pcmfm --waveform pcmfm --fuse 15 > gpurun_out/p_pcmfm.log 2>&1
echo ALLDONE
