"""Build libvf_hip.so (gfx950) in-tree with hipcc.  ``python -m viewformer_amd.build [--force]``.

hipcc cross-compiles without a GPU; the resulting .so sits next to this file so that it
travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libvf_hip.so')
ARCH = 'gfx950'
# per-source extra flags.  -fno-slp-vectorize: the SLP vectoriser turns the fp32 prologue math into v_pk_*_f32, which costs MFMA issue slots
# beside the matrix instructions (MI355X guide, 'price of one filler'); measured +0.3 ... +0.9 % on the dominant kernel without it
EXTRA_FLAGS = {'conv3_halo_x3h': ['-fno-slp-vectorize']}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    deps.append(os.path.join(os.path.dirname(HERE), 'include', 'vf_hip.h'))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    stems = {os.path.basename(s)[:-4] for s in sources()}
    for f in os.listdir(os.path.join(HERE, 'build')):              # objects of sources that have left csrc/ must not linger
        if f.endswith('.o') and f[:-2] not in stems:
            os.remove(os.path.join(HERE, 'build', f))
    for src in sources():
        obj = os.path.join(HERE, 'build', os.path.basename(src)[:-4] + '.o')
        cmd = [hipcc, f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-c', src, '-o', obj,
               '-Wall', '-Wno-unused-function'] + EXTRA_FLAGS.get(os.path.basename(src)[:-4], [])
        if verbose:
            cmd.append('-Rpass-analysis=kernel-resource-usage')
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0 or verbose:
            sys.stderr.write(f'--- {os.path.basename(src)}\n{out}\n')
        failed |= p.returncode != 0
    if failed:
        raise RuntimeError('hipcc failed')
    cmd = [hipcc, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', LIB] + objs
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='-v' in sys.argv))
