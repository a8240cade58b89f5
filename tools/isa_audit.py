#!/usr/bin/env python3
"""Static audit of the gfx950 code objects inside libwavenet_mi355.so: per kernel VGPRs / AGPRs / SGPRs, scratch
(private segment) bytes, spills, static LDS and the resulting waves per SIMD.  Runs without a GPU.

    python tools/isa_audit.py [--md profiles/<tag>_isa_audit.md]
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, 'tacotron-2_amd', 'csrc', 'libwavenet_mi355.so')
LLVM = '/opt/rocm/lib/llvm/bin'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def code_objects(so_path, workdir):
    """Yield paths of the gfx950 ELF code objects bundled in the library's .hip_fatbin section."""
    fat = os.path.join(workdir, 'fat.bin')
    subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, so_path], check=True)
    data = open(fat, 'rb').read()
    n = 0
    for m in re.finditer(MAGIC, data):
        base = m.start()
        (count,) = struct.unpack_from('<Q', data, base + len(MAGIC))
        p = base + len(MAGIC) + 8
        for _ in range(count):
            off, size, tlen = struct.unpack_from('<QQQ', data, p)
            triple = data[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if 'gfx950' in triple and size:
                path = os.path.join(workdir, 'co%d.elf' % n)
                open(path, 'wb').write(data[base + off:base + off + size])
                n += 1
                yield path


def kernels_of(elf):
    """Parse the AMDGPU metadata note (YAML rendering by llvm-readelf) into a list of dicts."""
    txt = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', elf], capture_output=True, text=True, check=True).stdout
    out, cur = [], None
    for line in txt.splitlines():
        m = re.match(r'\s+(-\s+)?\.(\w+):\s*(.*)$', line)
        if not m:
            continue
        dash, key, val = m.groups()
        if dash is not None and key not in ('address_space', 'name', 'offset', 'size', 'value_kind', 'actual_access', 'access', 'is_const'):
            cur = {}
            out.append(cur)
        if cur is not None and key in ('agpr_count', 'vgpr_count', 'sgpr_count', 'private_segment_fixed_size', 'group_segment_fixed_size',
                                      'vgpr_spill_count', 'sgpr_spill_count', 'max_flat_workgroup_size', 'symbol', 'uses_dynamic_stack'):
            cur[key] = val.strip().strip("'")
    return [k for k in out if 'symbol' in k]


def demangle(names):
    r = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True)
    return r.stdout.splitlines()


def audit(so_path=SO):
    rows = []
    with tempfile.TemporaryDirectory() as wd:
        for elf in code_objects(so_path, wd):
            ks = kernels_of(elf)
            names = demangle([k['symbol'].replace('.kd', '') for k in ks])
            for k, n in zip(ks, names):
                v, a = int(k.get('vgpr_count', 0)), int(k.get('agpr_count', 0))
                wg = int(k.get('max_flat_workgroup_size', 0))
                # unified 512-entry VGPR file per SIMD lane on CDNA3/4; allocation granule 8
                tot = max(8, (v + 7) // 8 * 8)
                waves = min(8, 512 // tot)
                rows.append(dict(name=re.sub(r'^void ', '', n).split('(')[0], vgpr=v, agpr=a, sgpr=int(k.get('sgpr_count', 0)),
                                 scratch=int(k.get('private_segment_fixed_size', 0)), vspill=int(k.get('vgpr_spill_count', 0)),
                                 sspill=int(k.get('sgpr_spill_count', 0)), lds=int(k.get('group_segment_fixed_size', 0)), wg=wg, waves=waves,
                                 dyn_stack=k.get('uses_dynamic_stack', 'false') == 'true'))
    rows.sort(key=lambda r: r['name'])
    return rows


def unwaited_load_hazards(so_path=SO, kernel_substr='synth_pipe_kernel'):
    """csrc/wn_synth_pipe.hip issues ring-tap prefetches as inline-asm `global_load_dwordx4 ... sc1` WITHOUT a wait
    (ld_g16_nowait) and waits by hand after the skip chain (pf_wait).  hipcc's waitcnt pass does not see loads inside inline asm: if
    register allocation ever copies, spills or re-uses the destination registers in that window the copy reads them before the data has
    landed.  This scans the disassembly of every kernel whose name contains `kernel_substr`: for each `global_load_dwordx4 ... sc1` that is
    not followed by its `s_waitcnt vmcnt(0)` within two instructions, every instruction up to the next `s_waitcnt vmcnt(0)` (program
    order) must leave the destination registers alone and must not be a scratch access.  Returns a list of (kernel, load, offender)."""
    bad, seen = [], 0
    with tempfile.TemporaryDirectory() as wd:
        for elf in code_objects(so_path, wd):
            txt = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--no-show-raw-insn', elf], capture_output=True, text=True, check=True).stdout
            cur, body = None, []
            blocks = []
            for line in txt.splitlines():
                m = re.match(r'^[0-9a-f]+ <(\S+)>:', line)
                if m:
                    cur = m.group(1); body = []; blocks.append((cur, body)); continue
                if cur is not None and line.strip() and not line.strip().startswith('//'):
                    body.append(line.split('//')[0].strip())
            for name, ins in blocks:
                if kernel_substr not in name or name.endswith('.kd'):
                    continue
                for i, x in enumerate(ins):
                    m = re.match(r'global_load_dwordx4 v\[(\d+):(\d+)\], .* sc1', x)
                    if not m or any(re.match(r's_waitcnt vmcnt\(0\)', y) for y in ins[i + 1:i + 3]):
                        continue
                    seen += 1
                    lo, hi = int(m.group(1)), int(m.group(2))
                    for y in ins[i + 1:]:
                        if re.match(r's_waitcnt vmcnt\(0\)', y):
                            break
                        if y.startswith('scratch_'):
                            bad.append((name, x, y)); continue
                        if re.match(r'global_load_dwordx4 v\[\d+:\d+\], .* sc1', y):      # a sibling prefetch: its own destination
                            m2 = re.match(r'global_load_dwordx4 v\[(\d+):(\d+)\]', y)
                            if not (int(m2.group(2)) < lo or int(m2.group(1)) > hi):
                                bad.append((name, x, y))
                            continue
                        regs = set()
                        for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', y):
                            regs.update(range(int(a), int(b) + 1))
                        regs.update(int(a) for a in re.findall(r'\bv(\d+)\b', y))
                        if any(lo <= r <= hi for r in regs):
                            bad.append((name, x, y))
    return seen, bad


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--unwaited':
        seen, bad = unwaited_load_hazards()
        print('%d un-waited sc1 loads, %d instructions touching their destination before the wait' % (seen, len(bad)))
        for b in bad:
            print('  ', b)
        return 1 if bad else 0
    rows = audit()
    lines = ['| kernel | VGPR (incl. AGPR) | AGPR | SGPR | scratch B | VGPR spills | static LDS B | max WG | waves/SIMD by VGPR |', '|---|---|---|---|---|---|---|---|---|']
    for r in rows:
        lines.append('| `%s` | %d | %d | %d | %d | %d | %d | %d | %d |' % (r['name'][:90], r['vgpr'], r['agpr'], r['sgpr'], r['scratch'], r['vspill'], r['lds'], r['wg'], r['waves']))
    text = '\n'.join(lines)
    if len(sys.argv) > 2 and sys.argv[1] == '--md':
        with open(sys.argv[2], 'w') as f:
            f.write('# Static ISA audit of libwavenet_mi355.so (gfx950), from the AMDGPU metadata notes\n\n`python tools/isa_audit.py --md %s`\n\n' % sys.argv[2])
            f.write('%d kernels; %d with scratch; %d with VGPR spills.  Dynamic LDS (the tile engine\'s ring) is requested at launch and not shown.\n\n' %
                    (len(rows), sum(r['scratch'] > 0 for r in rows), sum(r['vspill'] > 0 for r in rows)))
            f.write(text + '\n')
    else:
        print(text)
    return 0


if __name__ == '__main__':
    sys.exit(main())
