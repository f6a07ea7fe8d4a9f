cd /root/repo
cp pyg_lib_amd/libpyg_hip.so pyg_lib_amd/libpyg_hip_rb64.so
for v in rb64 rb32 rb16; do cp pyg_lib_amd/libpyg_hip_$v.so pyg_lib_amd/libpyg_hip.so; echo "== $v"; timeout 200 python tools/narrow_row_kernels.py 2>&1 | grep -v amdgpu; done
