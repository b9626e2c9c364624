"""The package's runtime switches, read from the environment ONCE, at import, into a frozen object.

Every `DPN_*` variable that steers the product path is listed here with its default; nothing else in the package reads the environment
(the C library reads three of its own, listed under LIBRARY_KNOBS: they select between kernels that are bit-identical or are timing
switches).  `bench.py` dumps `snapshot()` into its JSON line, so a measurement names the configuration it ran.
Tests and tools that compare two settings inside one process use `override(...)` -- an explicit, scoped replacement -- instead of writing
to `os.environ` behind the package's back.
"""
import contextlib
import dataclasses
import os


@dataclasses.dataclass(frozen=True)
class Config:
    # --- configurations named by BASELINE.json
    encoder_fp8: str = ''              # DPN_ENCODER_FP8 = 1 | mx : configs[4], the encoder layers' forward GEMMs on fp8 MFMA (off: it costs parity)
    # --- A/B switches (measurements in DESIGN.md are taken with them; the defaults are the product)
    encoder_unfused: bool = False      # DPN_ENCODER_UNFUSED=1 : round 3's per-GEMM encoder nodes instead of the row-local fused launches
    attn_fwd_fp32: bool = False        # DPN_ATTN_FWD=fp32 : the exact-fp32 attention forward of round 3 instead of dpn_attn16_fwd
    heads_per_field: str = '0'         # DPN_HEADS_PER_FIELD = 1 | fwd | bwd : per-field hyper-network head launches for lead batches
    heads_dmeta_parts: int = 6         # DPN_HEADS_DMETA_PARTS : side-by-side partial problems of the heads' input gradient
    enc_row_tiles: int = 0             # DPN_ENC_ROW_TILES : 16-row tiles per workgroup of dpn_enc_fwd / bwd (0: by problem size)
    embed_own_wgrad: bool = False      # DPN_EMBED_OWN_WGRAD=1 : the token convolution's weight gradient in its own launch
    embed_defer: bool = True           # DPN_EMBED_DEFER=0 : dpn_embed_assemble as a launch of its own (rounds 1-5) instead of inside the stack's first launch
    batch_eager_backward: bool = True  # DPN_BATCH_EAGER_BACKWARD=0 : park every field's saved state until the backward pass (lead batches)
    batch_pack: bool = True            # DPN_BATCH_PACK=0 : lead batches pack each field's weights and finish its losses in launches of their own (rounds 3-5)
    branches: tuple = ()               # DPN_BRANCHES = comma list of the side branches taken (branch.py): finish (the static half of the point
                                       # backward's finish stage), wgrad16 (the encoder's weight gradients above the first layer).  Default: none --
                                       # measured, every fork / join pair costs the captured step more than the branch hides
    # --- shelved experiments (kernels live in the experiment library, tools/variant_build.py; the product library refuses them)
    embed_gemm16: bool = False         # DPN_EMBED_GEMM16=1
    embed_parts: int = 0               # DPN_EMBED_PARTS
    embed_align: bool = True           # DPN_EMBED_ALIGN=0 : K-slices of the token convolution NOT rounded to whole 64-deep k-tiles (rounds 1-5: sixteen slices of 451)
    conv16: bool = False               # DPN_CONV16=1

    @staticmethod
    def from_env(env=None):
        e = os.environ if env is None else env
        fp8 = e.get('DPN_ENCODER_FP8', '')
        return Config(
            encoder_fp8=fp8 if fp8 in ('1', 'mx') else '',
            encoder_unfused=e.get('DPN_ENCODER_UNFUSED') == '1',
            attn_fwd_fp32=e.get('DPN_ATTN_FWD') == 'fp32',
            heads_per_field=e.get('DPN_HEADS_PER_FIELD', '0'),
            heads_dmeta_parts=int(e.get('DPN_HEADS_DMETA_PARTS', '6')),
            enc_row_tiles=int(e.get('DPN_ENC_ROW_TILES', '0')),
            embed_own_wgrad=e.get('DPN_EMBED_OWN_WGRAD') == '1',
            embed_defer=e.get('DPN_EMBED_DEFER', '1') == '1',
            batch_eager_backward=e.get('DPN_BATCH_EAGER_BACKWARD', '1') == '1',
            batch_pack=e.get('DPN_BATCH_PACK', '1') == '1',
            branches=tuple(b for b in e.get('DPN_BRANCHES', '').split(',') if b) if e.get('DPN_NO_BRANCHES') != '1' else (),
            embed_gemm16=e.get('DPN_EMBED_GEMM16') == '1',
            embed_parts=int(e.get('DPN_EMBED_PARTS', '0')),
            embed_align=e.get('DPN_EMBED_ALIGN', '1') != '0',
            conv16=e.get('DPN_CONV16') == '1',
        )


# read by the C library itself (csrc): DPN_FWD_KERNEL / DPN_BWD_KERNEL = ring | tiles pick between two bit-identical decompositions of the point
# kernels (tests compare them), DPN_ENC_NO_HELPERS drops the L2 warm-up workgroups of the encoder launches (a timing switch),
# DPN_BWD_ORDER / DPN_WGRAD_ORDER = reverse are the cache-residency probes of DESIGN.md section 4c (timing switches, off in the product)
LIBRARY_KNOBS = ('DPN_FWD_KERNEL', 'DPN_FWD_PP', 'DPN_FWD_PERSIST', 'DPN_SGEMM_TILE', 'DPN_ATTN_BWD_ROLES', 'DPN_BWD_KERNEL', 'DPN_ENC_NO_HELPERS', 'DPN_LIB', 'DPN_BWD_ORDER', 'DPN_WGRAD_ORDER')

FROZEN = Config.from_env()


def snapshot():
    """What a measurement ran with: the frozen switches that differ from their defaults, and every DPN_* variable present in the environment."""
    default = Config()
    changed = {f.name: getattr(FROZEN, f.name) for f in dataclasses.fields(Config) if getattr(FROZEN, f.name) != getattr(default, f.name)}
    return {'non_default': changed, 'environment': {k: v for k, v in sorted(os.environ.items()) if k.startswith('DPN_')}}


@contextlib.contextmanager
def override(**kw):
    """Scoped replacement of the frozen switches (tests / measurement tools only)."""
    global FROZEN
    old = FROZEN
    FROZEN = dataclasses.replace(old, **kw)
    try:
        yield FROZEN
    finally:
        FROZEN = old


def set_switches(**kw):
    """Unscoped replacement (measurement tools that walk through several settings in one process)."""
    global FROZEN
    FROZEN = dataclasses.replace(FROZEN, **kw)
    return FROZEN
