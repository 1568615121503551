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


# side-by-side builds of the same ABI that tests load next to the product library (never the product): name -> (extra flags, source stems)
VARIANTS = {
    # the transposing LDS reads of the DMA-ring kernels through the compiler intrinsic again instead of inline asm (vf_common.h: vf_tr_frag2_wait):
    # hipcc then orders them behind the ring with its own `s_waitcnt vmcnt(0)` — the slow but compiler-ordered reference of
    # tests/test_hip_ring_stress.py, which compares the two builds' results bit for bit
    'trintrin': (['-DVF_X_TRINTRIN'], ['attention_dma', 'attention_train_bf16', 'gemm_tn_bf16']),
    # A/B only (tools/ab_inprocess_*attn.py): the attention launches' owner blocks in index order instead of heaviest first
    'attn_index_order': (['-DADMA_HEAVY_FIRST=0', '-DATB_HEAVY_FIRST=0'], ['attention_dma', 'attention_train_bf16']),
    # A/B only: the x3h16 convolution's staging transform (GroupNorm-apply + swish + fp16 split) on packed fp32 instructions (bit-identical, slower)
    'x3h16_pk_xform': (['-DX3H16_PKXFORM=1'], ['conv3_halo_x3h']),
    # A/B only: rows per block of the LayerNorm backward (16 shipped)
    'ln_bwd_rpb8': (['-DVF_LN_BWD_RPB=8'], ['train_ops']),
    'ln_bwd_rpb32': (['-DVF_LN_BWD_RPB=32'], ['train_ops']),
    # A/B only: the dQ kernel with a two-slot ring (48 KB: three workgroups per CU instead of two, one tile ahead instead of two)
    'dq_ring2': (['-DATB_DQ_RING=2'], ['attention_train_bf16']),
    # A/B only: the dK / dV kernel's dropout words hashed by every lane (one per score) instead of once per lane quad (same words, same masks)
    # ablations of the backward attention kernels (results WRONG; timing only): no "tr" image DMAs / no tile compute / no tile DMAs
    'atb_abl1': (['-DATB_ABL=1'], ['attention_train_bf16']),
    'atb_abl2': (['-DATB_ABL=2'], ['attention_train_bf16']),
    'atb_abl4': (['-DATB_ABL=4'], ['attention_train_bf16']),
    'atb_abl9': (['-DATB_ABL=9'], ['attention_train_bf16']),
    'atb_abl6': (['-DATB_ABL=6'], ['attention_train_bf16']),
    'atb_abl6': (['-DATB_ABL=6'], ['attention_train_bf16']),
    # A/B only: the dK / dV kernel's mask rotation as shl / shr / or (before the third session of round 6)
    'dkv_rot3': (['-DVF_X_DKV_ROT3'], ['attention_train_bf16']),
    # A/B only: the dK / dV kernel with a rows image AND a tr image per streamed operand, ring of 2, two workgroups per CU (before the third session of round 6)
    'dkv_two_images': (['-DATB_KV_UNI=0'], ['attention_train_bf16']),
    # A/B only: the dQ kernel with K rows | V rows | K tr per slot (72 KB ring, two workgroups per CU)
    'dq_three_images': (['-DATB_DQ_UNI=0'], ['attention_train_bf16']),
    'dq_ring4': (['-DATB_DQ_RING=4'], ['attention_train_bf16']),
    # A/B only: both backward attention kernels as before the third session of round 6 (separate rows / tr images, two workgroups per CU, shl / shr / or rotation)
    'atb_session2': (['-DATB_KV_UNI=0', '-DATB_DQ_UNI=0', '-DVF_X_DKV_ROT3'], ['attention_train_bf16']),
    # A/B only: the forward DMA-ring attention with a three-slot ring (48 KB: three workgroups per CU, two tiles in flight)
    'adma_ring3': (['-DADMA_RING=3'], ['attention_dma']),
    'dkv_sel2': (['-DVF_X_DKV_SEL2'], ['attention_train_bf16']),
    # A/B only: the forward DMA-ring attention with four CONSECUTIVE query views per workgroup under the streams mask too
    'adma_consecutive': (['-DADMA_REGROUP=0'], ['attention_dma']),
    # A/B only: the backward attention kernels' tile lists from loops over visible() (integer divisions) instead of closed-form bit masks
    'atb_visloop': (['-DVF_X_ATB_VISLOOP'], ['attention_train_bf16']),
    # ablations of the forward DMA-ring kernel (results WRONG; timing only)
    'adma_nocompute': (['-DADMA_X_NOCOMPUTE'], ['attention_dma']),
    'adma_skeleton': (['-DADMA_X_NOCOMPUTE', '-DADMA_X_NODMA'], ['attention_dma']),
    'dkv_hash_per_element': (['-DVF_X_DKV_HASH_PER_ELEMENT'], ['attention_train_bf16']),
}


def variant_path(name: str) -> str:
    return os.path.join(HERE, 'variants', f'libvf_{name}.so')


def build_variant(name: str, force: bool = False) -> str:
    """libvf_<name>.so under viewformer_amd/variants/: the product objects, with the variant's sources recompiled under its flags"""
    flags, stems = VARIANTS[name]
    build()
    out = variant_path(name)
    srcs = [os.path.join(CSRC, st + '.hip') for st in stems]
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in srcs + hdrs + [LIB]):
        return out
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(os.path.dirname(out), exist_ok=True)
    procs, vobjs = [], []
    for st, src in zip(stems, srcs):
        obj = os.path.join(HERE, 'variants', f'{st}.{name}.o')
        cmd = [hipcc, f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-w', '-c', src, '-o', obj] + EXTRA_FLAGS.get(st, []) + flags
        procs.append(subprocess.Popen(cmd))
        vobjs.append(obj)
    if any(p.wait() != 0 for p in procs):
        raise RuntimeError(f'hipcc failed on variant {name}')
    objs = [os.path.join(HERE, 'build', os.path.basename(s)[:-4] + '.o') for s in sources() if os.path.basename(s)[:-4] not in stems]
    subprocess.check_call([hipcc, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', out] + objs + vobjs)
    return out


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='-v' in sys.argv))
