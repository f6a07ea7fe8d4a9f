#!/usr/bin/env python3
"""Kernel resource table of one HIP translation unit: python tools/kres.py hip/matmul_dw.hip [extra hipcc flags]
(name, VGPRs, AGPRs, SGPRs, scratch bytes / lane, LDS bytes, occupancy) from -Rpass-analysis=kernel-resource-usage."""
import re, subprocess, sys, os
src = sys.argv[1]
here = os.path.dirname(os.path.abspath(__file__))
csrc = os.path.join(here, '..', 'pyg_lib_amd', 'csrc')
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-munsafe-fp-atomics', '-fPIC',
       '-I' + os.path.join(here, '..', 'include'), '-I' + os.path.join(csrc, 'hip'), '-c', src, '-o', '/dev/null',
       '-Rpass-analysis=kernel-resource-usage'] + sys.argv[2:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for ln in err.splitlines():
    m = re.search(r'remark: +([A-Za-z ]+?)(?: \[bytes/(?:lane|block)\])?: (.+?) \[-Rpass', ln)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == 'Function Name':
        if cur:
            rows.append(cur)
        cur = {'name': subprocess.run(['c++filt', v], capture_output=True, text=True).stdout.strip()}
    else:
        cur[k] = v
if cur:
    rows.append(cur)
for r in rows:
    n = re.sub(r'pyg_hip::\(anonymous namespace\)::', '', r['name'])
    n = re.sub(r'\(.*', '', n)[:70]
    print(f"{n:70s} v{r.get('VGPRs','?'):>4} a{r.get('AGPRs','?'):>4} s{r.get('SGPRs','?'):>4} scratch {r.get('ScratchSize','?'):>5} "
          f"lds {r.get('LDS Size','?'):>6} occ {r.get('Occupancy','?')}")
