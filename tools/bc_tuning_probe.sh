#!/bin/bash
# ConvLSTM-shape bottleneck convolution in the tools-only build: request burst (CH) x wave target (splits)
cd "$(dirname "$0")/.."
export DVMVS_HIP_LIB=$PWD/deep-video-mvs_amd/lib/libdvmvs_hip_tuning.so
for waves in 1024 2048 4096; do for ch in 0 1 2; do
  echo -n "waves $waves CH $ch: "; DVMVS_BC_WAVES=$waves DVMVS_BC_CH=$ch timeout 60 python tools/lstm_conv_probe.py --kernel-only 2>/dev/null | grep "1024->2048"
done; done
