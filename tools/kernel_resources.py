"""Register / spill / scratch / occupancy table of the library's kernels, from hipcc's own remarks.

    python tools/kernel_resources.py [-DNAME=VALUE ...] [--filter substring]

Compiles neuray_amd/csrc/neuray_hip.hip for gfx950 with -Rpass-analysis=kernel-resource-usage (device code only, no link) and
prints one line per kernel: VGPRs, AGPRs, spilled VGPRs / SGPRs, scratch bytes per lane, occupancy (waves per SIMD), LDS bytes."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neuray_amd import build as nbuild  # noqa: E402


def resources(extra=()):
    flags = [f for f in nbuild.FLAGS if f not in ('-shared', '-fPIC')]
    cmd = [nbuild.HIPCC] + flags + list(extra) + ['-Rpass-analysis=kernel-resource-usage', '--cuda-device-only', '-c',
                                                   nbuild.SOURCES[0], '-o', '/dev/null']
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    out, cur = [], None
    for line in err.splitlines():
        m = re.search(r'remark: .*Function Name: (\S+)', line)
        if m:
            cur = {'name': subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()}
            out.append(cur)
            continue
        m = re.search(r'remark: .*?\s{2,}([A-Za-z ]+(?:\[[^\]]*\])?[A-Za-z ]*): (\d+)', line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


def main():
    extra = [a for a in sys.argv[1:] if a.startswith('-D')]
    filt = sys.argv[sys.argv.index('--filter') + 1] if '--filter' in sys.argv else ''
    rows = [r for r in resources(extra) if filt in r['name']]
    print('%-100s %5s %5s %6s %6s %8s %4s %7s' % ('kernel', 'VGPR', 'AGPR', 'vspill', 'sspill', 'scratch', 'occ', 'LDS'))
    for r in rows:
        name = re.sub(r'^void ', '', r['name'])
        name = re.sub(r'\(.*$', '', name)
        print('%-100s %5d %5d %6d %6d %8d %4d %7d' % (name[:100], r.get('VGPRs', -1), r.get('AGPRs', -1), r.get('VGPRs Spill', -1),
                                                       r.get('SGPRs Spill', -1), r.get('ScratchSize [bytes/lane]', -1),
                                                       r.get('Occupancy [waves/SIMD]', -1), r.get('LDS Size [bytes/block]', -1)))


if __name__ == '__main__':
    main()
