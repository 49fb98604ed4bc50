cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in "" _precise _ptrw _ptrm; do
  echo "==== variant '$v'" >> gpurun_out/diag3.log
  NEURAY_HIP_LIB=$GRAFT_REPO_ROOT/neuray_amd/libneuray_hip$v.so timeout 300 python gpu_diag2.py 2>&1 | grep -E "mu0|hit |geo |pixel|Error|error" >> gpurun_out/diag3.log
done
cat gpurun_out/diag3.log
