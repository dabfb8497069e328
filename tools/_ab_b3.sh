python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "philox or awgn or noise or box_muller or oracle_1e7 or link" 2>&1 | tail -2
bash tools/ab_trees.sh _ab_head 3 --ber-points none
bash tools/ab_trees.sh _ab_head 2 --detector PAM --ber-points none
bash tools/ab_trees.sh _ab_head 2 --sps 10 --ber-points none
bash tools/ab_trees.sh _ab_head 2 --waveform pcmfm --ber-points none
