#!/usr/bin/env python
"""
Register / spill / scratch report of every gfx950 kernel of the engine, from the compiler itself
(hipcc -Rpass-analysis=kernel-resource-usage): one CSV row per kernel instantiation.

    python tools/kernel_resources.py > profiles/r03_kernel_resources.csv        (no GPU needed; ~4 minutes, 8 jobs)

Columns: file, kernel (demangled), vgprs, agprs, sgprs, sgpr_spills, vgpr_spills, scratch_bytes_per_lane,
occupancy_waves_per_simd, static_lds_bytes (the dynamic LDS is chosen at launch, bsx_api.cpp).
`launched_by` marks the instantiations the BASELINE configs and the bench launch (tools/bench_configs.py).
"""
import concurrent.futures
import csv
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'boolsi_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
CXXFILT = 'c++filt'

# what the measured configurations launch (kernel name prefix -> who)
LAUNCHED = {
    'bsx::k_attract_pool<2, 2, 1, true, false>': 'bench / north star (n = 64, K = 2): top level of the cube cascade (the dominant launch)',
    'bsx::k_attract_pool<2, 2, 1, true, true>': 'bench / north star: lower levels of the cascade',
    'bsx::k_attract_pool<2, 2, 1, false, false>': 'north star: ragged ends, plain tiles',
    'bsx::k_attract<2, 2, 1>': 'north star / config 4: discovery, unresolved classes',
    'bsx::k_attract_pool<1, 2, 1, true, false>': 'config 3 (n = 32, K = 2): cube cascade, top level',
    'bsx::k_attract_pool<1, 2, 1, true, true>': 'config 3: cube cascade, lower levels',
    'bsx::k_attract_pool<1, 4, 1, true, false>': 'config 2: cambium2 (n = 30, k_mux = 4): one cube pass at depth 5',
    'bsx::k_attract<1, 4, 1>': 'config 2: cambium2 discovery',
    'bsx::k_attract<1, 2, 1>': 'config 3: discovery',
    'bsx::k_target<2, 2, 1>': 'config 4 (n = 64, K = 2 target)',
    'bsx::k_simulate_sliced64<4, 3, true>': 'config 5 (n = 128, K = 3 simulate) with digests',
    'bsx::k_simulate_sliced64<4, 3, false>': 'config 5, final states only',
    'bsx::k_attract<2, 3, 1>': 'chaotic K = 3 networks (n = 64): every trajectory through the detector',
    'bsx::k_compact_near': 'cube cascade: packing between levels',
    'bsx::k_publish': 'cube cascade: counters to the host',
}


def one(path):
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [HIPCC, '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, '--offload-arch=gfx950',
               '-Rpass-analysis=kernel-resource-usage', '-c', path, '-o', os.path.join(tmp, 'x.o')]
        if path.endswith('.cpp'):
            cmd[1:1] = ['-x', 'hip']
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        if 'remark' not in line:
            continue
        m = re.search(r'(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|'
                      r'SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)', line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == 'Function Name':
            cur = {'file': os.path.basename(path), 'mangled': v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    return rows


def main():
    files = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    with concurrent.futures.ThreadPoolExecutor(max_workers=int(os.environ.get('JOBS', '8'))) as ex:
        rows = [r for part in ex.map(one, files) for r in part]
    names = subprocess.run([CXXFILT], input='\n'.join(r['mangled'] for r in rows), capture_output=True, text=True).stdout.splitlines()
    w = csv.writer(sys.stdout)
    w.writerow(['file', 'kernel', 'vgprs', 'agprs', 'sgprs', 'sgpr_spills', 'vgpr_spills', 'scratch_bytes_per_lane',
                'occupancy_waves_per_simd', 'static_lds_bytes', 'launched_by'])
    for r, name in zip(rows, names):
        name = re.sub(r'\(.*\)$', '', name).replace('void ', '')
        who = ''
        for prefix, what in LAUNCHED.items():
            if name.startswith(prefix):
                who = what
        w.writerow([r['file'], name, r.get('VGPRs'), r.get('AGPRs'), r.get('TotalSGPRs'), r.get('SGPRs Spill'), r.get('VGPRs Spill'),
                    r.get('ScratchSize [bytes/lane]'), r.get('Occupancy [waves/SIMD]'), r.get('LDS Size [bytes/block]'), who])


if __name__ == '__main__':
    main()
