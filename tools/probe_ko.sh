# knock-out builds (csrc/Makefile `make ko KO=<bits>`) against the shipped library, a few layers x variants each: what does a piece of a kernel cost?
L="${PROBE_LAYERS:-backbone.8.m.0.conv2:-2,24,5152:10,20,32 backbone.6.m.0.conv2:-2,40,5136:8,20,64 backbone.31.cls_reg_conv:-2,40,2592:16,16,32 backbone.16.m.0.conv2:-2,40,5136}"
for k in "" ${PROBE_KO:-32 64 128 96}; do
  if [ -z "$k" ]; then python tools/conv_probe.py $L; else MAF_HIP_LIB=$PWD/maf-yolo_amd/libmafyolo_ko$k.so python tools/conv_probe.py $L; fi
done 2>&1 | grep -v amdgpu.ids
