for so in viewformer_amd/variants/libvf_*.so; do
  n=$(basename $so .so); n=${n#libvf_}
  VF_HIP_LIB=$PWD/$so python - <<PY 2>&1 | grep -v amdgpu | sed "s/^/[$n] /"
import sys; sys.path.insert(0,'.'); sys.argv=['x']
import tools.microbench as mb
mb.vqf_stamps(64*256)
mb.vqf_stamps(64*896)
PY
done
