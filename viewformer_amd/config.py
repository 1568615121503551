"""Model configuration dataclasses.

Field names and defaults mirror the reference's ``config.json`` contract
(reference: viewformer/models/config.py:63-88 ``MIGTConfig``, :92-119
``VQGANConfig``, viewformer/models/__init__.py:62-78 ``load_config``) so that a
``config.json`` written by the reference trainer loads here unchanged.
"""
from dataclasses import dataclass, field, fields, asdict
from typing import List


@dataclass
class VQGANConfig:
    learning_rate: float = 1.584e-3
    embed_dim: int = 256
    n_embed: int = 1024
    z_channels: int = 256
    resolution: int = 256
    in_channels: int = 3
    out_ch: int = 3
    ch: int = 128
    num_res_blocks: int = 2
    ch_mult: List[int] = field(default_factory=lambda: [1, 1, 2, 2, 4])
    attn_resolutions: List[int] = field(default_factory=lambda: [16])
    gradient_clip_val: float = .0
    batch_size: int = 352
    image_size: int = 128
    total_steps: int = 200000
    codebook_weight: float = 1.0
    pixelloss_weight: float = 1.0
    perceptual_weight: float = 1.0
    model: str = 'vqgan'

    @property
    def stride(self) -> int:
        return 2 ** (len(self.ch_mult) - 1)

    @property
    def model_type(self) -> str:
        return 'codebook'

    def asdict(self):
        return asdict(self)


@dataclass
class MIGTConfig:
    n_embeddings: int = 1024
    n_head: int = 12
    d_model: int = 768
    dropout: float = 0.1
    n_layer: int = 12
    weight_decay: float = 0.01
    label_smoothing: float = 0.0
    learning_rate: float = 6.4e-4
    batch_size: int = 64
    gradient_clip_val: float = 0.0
    sequence_size: int = 20
    token_image_size: int = 8
    total_steps: int = 300000
    n_loss_skip: int = 4
    augment_poses: str = 'relative'   # 'no' | 'relative' | 'simple' | 'advanced'
    use_dynamic_pose_loss: bool = False
    localization_weight: str = '1'    # schedule string; '0' disables the localization head
    image_generation_weight: float = 1.
    pose_multiplier: float = 1.
    random_pose_multiplier: float = 1.
    model: str = 'migt'

    @property
    def model_type(self) -> str:
        return 'transformer'

    @property
    def use_localization(self) -> bool:
        """Reference: migt.py:268-269 (``not localization_weight.is_zero()``)."""
        from .schedules import parse
        return not parse(self.localization_weight).with_total_steps(self.total_steps).is_zero()

    def asdict(self):
        return asdict(self)


_CONFIGS = {'vqgan': VQGANConfig, 'migt': MIGTConfig}


def config_field_names(model_name: str):
    """Field names of the config class a ``config.json`` with ``model == model_name`` builds."""
    if model_name not in _CONFIGS:
        raise ValueError(f'Model {model_name} is not supported')
    return {f.name for f in fields(_CONFIGS[model_name])}


def load_config(config: dict):
    """Build a config object from a ``config.json`` dict (reference:
    viewformer/models/__init__.py:62-78).  Unknown model names raise ``ValueError``; keys that are not fields of the
    config class are ignored, as the reference's ``_build_dataclass`` (:63-74) does: it walks the dataclass fields and
    never looks at the rest, so a ``config.json`` the reference accepts loads here too.  Schedule-typed fields stay
    text (``str(Schedule)`` is what ``asdict`` writes, config.py:14-15)."""
    config = dict(config)
    name = config.pop('model')
    if name not in _CONFIGS:
        raise ValueError(f'Model {name} is not supported')
    cls = _CONFIGS[name]
    known = {f.name for f in fields(cls)}
    kw = {k: v for k, v in config.items() if k in known}
    if 'localization_weight' in kw:
        kw['localization_weight'] = str(kw['localization_weight'])
    return cls(**kw)
