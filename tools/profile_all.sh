set -e
bash tools/profile.sh r03_fuse15 --fuse 15 > gpurun_out/p_fuse15.log 2>&1
bash tools/profile.sh r03_sps10 --sps 10 --fuse 15 > gpurun_out/p_sps10.log 2>&1
bash tools/profile.sh r03_multih --waveform multih --fuse 15 > gpurun_out/p_multih.log 2>&1
bash tools/profile.sh r03_pcmfm --waveform pcmfm --fuse 15 > gpurun_out/p_pcmfm.log 2>&1
echo ALLDONE
