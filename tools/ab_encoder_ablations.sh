cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in hip nodma nomfma both; do
  if [ $v = hip ]; then unset NEUBE_LIB_PATH; else export NEUBE_LIB_PATH=$PWD/brushstroke_engine_amd/csrc/libneube_$v.so; fi
  echo "== $v"
  rocprofv3 --kernel-trace -d gpurun_out/enctrace -o enc --output-format csv -- python3 tools/trace_encoder.py 2>&1 | grep "^encoder"
  python3 tools/trace_encoder_summary.py gpurun_out/enctrace | grep "enc_conv3x3_h3_kernel<2"; rm -rf gpurun_out/enctrace
done
