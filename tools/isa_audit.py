"""Static audit of the gfx950 code objects: per kernel the register budget
hipcc settled on (VGPR / AGPR / spills / waves per SIMD) and, for kernels with
MFMA loops, how often an MFMA waits on an LDS read issued right in front of it
(`ds_read; s_waitcnt lgkmcnt(0); v_mfma`: one exposed LDS latency per group —
the pattern the weight-gradient kernels were stuck in).

    python tools/isa_audit.py [file.hip ...] > profiles/rNN_isa_audit.txt

Runs on the build box (no GPU): hipcc -S per source."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'xrdslam_amd', 'csrc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
         '-munsafe-fp-atomics', '-fno-gpu-rdc', '-I' + os.path.join(ROOT, 'include'),
         '-I' + CSRC, '-x', 'hip', '-S', '--cuda-device-only',
         '-Rpass-analysis=kernel-resource-usage']


def demangle(names):
    try:
        out = subprocess.run(['c++filt'] + names, capture_output=True,
                             text=True, check=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def audit(src):
    asm = '/tmp/isa_audit_%s.s' % os.path.basename(src)
    p = subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + [src, '-o', asm],
                       capture_output=True, text=True)
    res, name = {}, None
    for line in p.stderr.splitlines():
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            name = m.group(1)
            res[name] = {}
            continue
        for key, tag in (('VGPRs:', 'vgpr'), ('AGPRs:', 'agpr'),
                         ('VGPRs Spill:', 'spill'),
                         ('Occupancy [waves/SIMD]:', 'occ'),
                         ('ScratchSize [bytes/lane]:', 'scratch')):
            if name and key in line:
                res[name][tag] = int(line.split(key)[1].split()[0])
    # per-function instruction streams
    body, cur = {}, None
    for line in open(asm):
        m = re.match(r'^(_Z\w+):', line)
        if m and m.group(1) in res:
            cur = m.group(1)
            body[cur] = []
            continue
        if line.startswith('\t.end_amdhsa_kernel') or \
                line.startswith('.Lfunc_end'):
            cur = None
        t = line.strip()
        if cur and t and not t.startswith((';', '.')):
            body[cur].append(t.split()[0] + ' ' + ' '.join(t.split()[1:3]))
    for fn, ins in body.items():
        mfma = sum(i.startswith('v_mfma') for i in ins)
        exposed = 0
        for k, i in enumerate(ins):
            if i.startswith('s_waitcnt') and 'lgkmcnt(0)' in i and \
                    k + 1 < len(ins) and ins[k + 1].startswith('v_mfma') and \
                    any(x.startswith('ds_read') for x in ins[max(0, k - 3):k]):
                exposed += 1
        res[fn].update(mfma=mfma, exposed=exposed)
    return res


def main():
    srcs = sys.argv[1:] or sorted(
        os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))
    print('# kernel | VGPR AGPR spill waves/SIMD | MFMA instructions | '
          'MFMAs that wait on an LDS read issued <= 3 instructions earlier')
    for src in srcs:
        res = audit(src)
        names = demangle(list(res))
        print('## ' + os.path.relpath(src, ROOT))
        for fn, r in sorted(res.items(), key=lambda kv: -kv[1].get('mfma', 0)):
            short = names[fn].replace('xrd::(anonymous namespace)::', '')
            short = re.sub(r'\(.*', '', short)[:64]
            flag = ' <-- spills' if r.get('spill', 0) else ''
            if r.get('mfma') and r['exposed'] * 4 > r['mfma']:
                flag += ' <-- serialised LDS reads'
            print(f"{short:64s} | {r.get('vgpr', 0):3d} {r.get('agpr', 0):3d} "
                  f"{r.get('spill', 0):3d} {r.get('occ', 0):2d} | "
                  f"{r.get('mfma', 0):5d} | {r.get('exposed', 0):4d}{flag}")


if __name__ == '__main__':
    main()
