"""Single-/dual-modality ablations -- reference ``models/ablation_models.py``."""
from ._core import MultimodalNet, Spec

__all__ = ["SequenceModel", "SequenceFpModel", "StructureModel", "StructureModel_SSL", "StructureModelv2",
           "DualModel"]


class _Ablation(MultimodalNet):
    def __init__(self, vae_input_dim, device, gcn_layers=5, vae_hidden_dim=512, vae_latent_dim=32,
                 gat_hidden_channels=64, *args, mlp_features=32, **kwargs):
        super().__init__(vae_input_dim, device, gcn_layers, vae_hidden_dim, vae_latent_dim, gat_hidden_channels,
                         mlp_features=mlp_features)


class SequenceModel(_Ablation):
    """sequence VAE only -- reference ``ablation_models.py:10-66``"""
    SPEC = Spec(graph=False, prop="none")


class SequenceFpModel(_Ablation):
    """sequence VAE + raw 2-d property -- reference ``ablation_models.py:68-125``"""
    SPEC = Spec(graph=False, prop="raw")


class StructureModel(_Ablation):
    """graph only, 8-head node attention (the default ``--model`` of train_IEDB_wFT.py:17) -- ``:127-180``"""
    SPEC = Spec(attn="mha8", vae=False, prop="none")


class StructureModel_SSL(_Ablation):
    """reference ``ablation_models.py:182-242``"""
    SPEC = Spec(attn="mha8", vae=False, prop="none", ssl=True)


class StructureModelv2(_Ablation):
    """mean || max readout -- reference ``ablation_models.py:244-307``"""
    SPEC = Spec(attn="mha8", vae=False, prop="none", ssl=True, pool="meanmax")


class DualModel(MultimodalNet):
    """graph + sequence, no property branch -- reference ``ablation_models.py:309-398``"""
    SPEC = Spec(attn="v1", prop="none")

    def __init__(self, vae_input_dim, device, gcn_layers=5, vae_hidden_dim=512, vae_latent_dim=32,
                 gat_hidden_channels=64):
        super().__init__(vae_input_dim, device, gcn_layers, vae_hidden_dim, vae_latent_dim, gat_hidden_channels)
