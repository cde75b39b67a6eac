"""Builds libdvd_hip.so (gfx950) in-tree with hipcc.

    python -m dvd_hip.build [--force]

hipcc cross-compiles without a GPU.  Objects and the library live under
dvd_hip/lib/ (git-ignored; they travel to the GPU box with the tree).
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(ROOT, 'include')
LIBDIR = os.path.join(HERE, 'lib')
LIBNAME = 'libdvd_hip.so'
ARCH = 'gfx950'

# (source, extra flags).  The geometry kernels reproduce the reference's fp32
# operation order, so the compiler must not contract a*b+c behind our back.
SOURCES = [
    ('core.hip', []),
    ('unproject.hip', ['-ffp-contract=off']),
    ('warp_loss.hip', ['-ffp-contract=off']),
    ('warp_strip.hip', ['-ffp-contract=off']),
    ('sf_mlp.hip', []),
    ('elementwise.hip', ['-ffp-contract=off']),
    ('gconv.hip', []),
    ('gconv32.hip', []),
    ('surfaces.hip', ['-ffp-contract=off']),
    ('upsample.hip', ['-ffp-contract=off']),
    ('bnrelu.hip', []),
    ('amax.hip', []),
    ('xconv.hip', []),
    ('xwgrad.hip', []),
    ('xwgrad3.hip', []),
    ('consistency.hip', ['-ffp-contract=off']),
    ('a16.hip', []),
    ('pool.hip', []),
]
COMMON = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics',
          '-I' + INCLUDE, '-I' + CSRC]


def lib_path():
    return os.path.join(LIBDIR, LIBNAME)


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def _digest(paths, flags):
    h = hashlib.sha256()
    for p in paths:
        with open(p, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(flags).encode())
    return h.hexdigest()


def _local_includes(path, seen):
    """csrc/ headers a source includes, transitively (the public C ABI header is left out: it changes with every new entry
    point of ANY kernel and says nothing about the code of the units a profile is about)."""
    import re
    for name in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path).read(), re.M):
        h = os.path.join(CSRC, name)
        if os.path.exists(h) and h not in seen:
            seen.add(h)
            _local_includes(h, seen)
    return seen


def source_digest(units=None):
    """sha256 (16 hex digits) over the sources a set of translation units is built from -- the .hip files named in `units`
    (None: every unit of the library), the csrc/ headers they include, and their flags.  A profile that bench.py joins into its
    line (profiles/warp_loss_pmc.json, warp_loss_sq.json, mfma_roofline.json) is stamped with it by the tool that writes it on
    the GPU box; bench.py recomputes it and says whether the profile is of the kernels it is running."""
    h = hashlib.sha256()
    for src, extra in SOURCES:
        if units is not None and src not in units:
            continue
        sp = os.path.join(CSRC, src)
        # (flags without the include paths: they are absolute, and a profile collected in one checkout must match another)
        flags = [f for f in COMMON if not f.startswith('-I')] + extra
        h.update(_digest([sp] + sorted(_local_includes(sp, set())), flags).encode())
    return h.hexdigest()[:16]


WARP_UNITS = ('warp_loss.hip', 'warp_strip.hip')      # the fused warp+loss launch sequence


def build_library(force=False, verbose=False):
    """Compile every HIP translation unit and link the shared library.
    Returns the library path.  Re-uses objects whose sources did not change."""
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith('.h')]
    headers.append(os.path.join(INCLUDE, 'dvd_hip.h'))
    objs, jobs = [], []
    relink = force or not os.path.exists(lib_path())
    for src, extra in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(LIBDIR, src.replace('.hip', '.o'))
        stamp = obj + '.sha'
        dig = _digest([sp] + headers, COMMON + extra)
        fresh = os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig
        if force or not fresh:
            jobs.append((src, [_hipcc()] + COMMON + extra + ['-c', sp, '-o', obj], stamp, dig))
        objs.append(obj)

    def compile_one(job):
        src, cmd, stamp, dig = job
        if verbose:
            print(' '.join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
        with open(stamp, 'w') as f:
            f.write(dig)

    if jobs:                                    # translation units are independent: compile them concurrently
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
            list(pool.map(compile_one, jobs))
        relink = True
    if relink:
        cmd = [_hipcc(), '--offload-arch=' + ARCH, '-shared', '-fPIC'] + objs + ['-o', lib_path()]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
    return lib_path()


if __name__ == '__main__':
    print(build_library(force='--force' in sys.argv, verbose=True))
