set -e
bash tools/profile.sh r02_fuse15 > gpurun_out/p_fuse15.log 2>&1
bash tools/profile.sh r02_fuse7 --fuse 7 > gpurun_out/p_fuse7.log 2>&1
bash tools/profile.sh r02_unfused --fuse 0 > gpurun_out/p_unfused.log 2>&1
bash tools/profile.sh r02_multih --waveform multih > gpurun_out/p_multih.log 2>&1
bash tools/profile.sh r02_multih_fuse2 --waveform multih --fuse 2 > gpurun_out/p_multih2.log 2>&1
bash tools/profile.sh r02_pcmfm --waveform pcmfm > gpurun_out/p_pcmfm.log 2>&1
echo ALLDONE
