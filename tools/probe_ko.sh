# knock-out builds of the conv kernels (csrc/Makefile `make ko KO=<bits>`) against the shipped library, a few layers x variants each
L="${PROBE_LAYERS:-backbone.20.conv1:1,8,5:1,4,5:1,4,2 backbone.22.conv1:1,8,5:1,4,2 backbone.22.conv2:1,8,5:1,4,5:2,8,1 backbone.31.stem:1,8,5:1,4,3 backbone.4.conv2:1,6,5:2,6,1}"
for k in "" ${PROBE_KO:-1 2 3 4 8 16}; do
  if [ -z "$k" ]; then python tools/conv_probe.py $L; else MAF_HIP_LIB=$PWD/maf-yolo_amd/libmafyolo_ko$k.so python tools/conv_probe.py $L; fi
done 2>&1 | grep -v amdgpu.ids
