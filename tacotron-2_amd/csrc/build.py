#!/usr/bin/env python3
"""Build libwavenet_mi355.so for gfx950 with hipcc (cross-compiles without a GPU).

    python tacotron-2_amd/csrc/build.py [--force]

One hipcc invocation per translation unit (in parallel), then one link.  The .so is written next to
the sources (git-ignored, but it travels to the GPU box with the gpurun snapshot).
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ['wn_api.hip', 'wn_pack.hip', 'wn_frontend.hip', 'wn_loss.hip', 'wn_optim.hip', 'wn_train.hip', 'wn_synth.hip', 'wn_synth_pipe.hip', 'wn_f32.hip', 'wn_synth_f32.hip']
HEADERS = ['wn_common.h', 'wn_tile.h', 'wn_tile8p.h', 'wn_wgrad.h', 'wn_mulaw_tables.h', os.path.join('..', '..', 'include', 'wavenet_mi355.h')]
LIB = os.path.join(HERE, 'libwavenet_mi355.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++20', '-fPIC', '-munsafe-fp-atomics', '-Wall', '-Wno-unused-function', '-Wno-unused-value', '-Wno-inline-asm']


LAST_BUILD = {}      # translation unit -> 'compiled' | 'reused' (object newer than its source, every header and this script); 'link' likewise


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def _compile(src):
    obj = os.path.join(HERE, src.replace('.hip', '.o'))
    newest = max([_mtime(os.path.join(HERE, src))] + [_mtime(os.path.join(HERE, h)) for h in HEADERS] + [_mtime(__file__)])
    if _mtime(obj) > newest:
        LAST_BUILD[src] = 'reused'
        return obj, ''
    LAST_BUILD[src] = 'compiled'
    cmd = ['hipcc'] + FLAGS + ['-c', os.path.join(HERE, src), '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
    return obj, r.stderr


def build(force=False, verbose=True):
    if force:
        for s in SOURCES:
            o = os.path.join(HERE, s.replace('.hip', '.o'))
            if os.path.exists(o):
                os.remove(o)
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        res = list(ex.map(_compile, SOURCES))
    objs = [o for o, _ in res]
    warn = ''.join(w for _, w in res)
    if verbose and warn.strip():
        print(warn, file=sys.stderr)
    LAST_BUILD['link'] = 'reused'
    if _mtime(LIB) < max(_mtime(o) for o in objs):
        LAST_BUILD['link'] = 'linked'
        cmd = ['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
    return LIB


SAN_LIB = os.path.join(HERE, 'libwavenet_mi355_san.so')
SAN_FLAGS = ['-O1', '-g', '-fno-omit-frame-pointer', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-fno-gpu-sanitize']


def build_sanitized(verbose=True):
    """ASAN + UBSAN variant of the library's HOST side for the CPU ABI tests (tests/test_host_cpu.py runs them under it in a child
    process with the sanitizer runtime preloaded): wn_api.hip -- every C-ABI entry point, the configuration validation, the parameter
    table and workspace planning -- is rebuilt with -fsanitize=address,undefined (device code untouched: -fno-gpu-sanitize) and linked
    with the regular objects of the other translation units.  libwavenet_mi355_san.so; never loaded by the product."""
    build(verbose=verbose)
    src, obj = os.path.join(HERE, 'wn_api.hip'), os.path.join(HERE, 'wn_api.san.o')
    newest = max([_mtime(src)] + [_mtime(os.path.join(HERE, h)) for h in HEADERS] + [_mtime(__file__)])
    if _mtime(obj) <= newest:
        flags = [f for f in FLAGS if f != '-O3'] + SAN_FLAGS
        r = subprocess.run(['hipcc'] + flags + ['-c', src, '-o', obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc (sanitized) failed for wn_api.hip:\n%s\n%s' % (r.stdout, r.stderr))
    others = [os.path.join(HERE, s.replace('.hip', '.o')) for s in SOURCES if s != 'wn_api.hip']
    if _mtime(SAN_LIB) < max(_mtime(o) for o in others + [obj]):
        r = subprocess.run(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-fsanitize=address,undefined', '-shared-libsan', '-o', SAN_LIB, obj] + others, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link (sanitized) failed:\n%s\n%s' % (r.stdout, r.stderr))
    return SAN_LIB


def sanitizer_runtime():
    """Path of clang's shared ASAN runtime (to LD_PRELOAD into the python that loads libwavenet_mi355_san.so)."""
    r = subprocess.run(['/opt/rocm/lib/llvm/bin/clang', '-print-file-name=libclang_rt.asan-x86_64.so'], capture_output=True, text=True)
    p = r.stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


if __name__ == '__main__':
    if '--pipe-svc' in sys.argv:      # diagnostic build: the synthesis pipeline with its service-time stamps compiled in (tools/pipe_svc_trace.py)
        FLAGS.append('-DWN_PIPE_SVC_BUILD')
    if '--ablate' in sys.argv:        # diagnostic build: WN_ABLATE skips whole launch classes of the training step (wn_train.hip; results are wrong by construction)
        FLAGS.append('-DWN_ABLATE_BUILD')
        o = os.path.join(HERE, 'wn_train.o')
        if os.path.exists(o):
            os.remove(o)
    print(build_sanitized() if '--sanitize' in sys.argv else build(force='--force' in sys.argv or '--pipe-svc' in sys.argv))
    print('build mode: ' + ', '.join('%s %s' % kv for kv in sorted(LAST_BUILD.items())))
