"""Multimodal (structure + sequence + property) models -- reference ``models/hybrid_models.py``."""
from ._core import MultimodalNet, Spec

__all__ = ["HybridModel", "HybridModelv2", "HybridModel_SSL", "HybridModelv2_SSL"]


class HybridModel(MultimodalNet):
    """reference ``hybrid_models.py:10-119``"""
    SPEC = Spec(attn="v1")

    def __init__(self, vae_input_dim, device, gcn_layers: int = 5, vae_hidden_dim: int = 512,
                 vae_latent_dim: int = 32, gat_hidden_channels: int = 64, property_embedding_dim: int = 8,
                 *args, **kwargs):
        super().__init__(vae_input_dim, device, gcn_layers, vae_hidden_dim, vae_latent_dim, gat_hidden_channels,
                         property_embedding_dim=property_embedding_dim)


class HybridModel_SSL(MultimodalNet):
    """reference ``hybrid_models.py:121-238``"""
    SPEC = Spec(attn="v1", ssl=True)

    def __init__(self, vae_input_dim, device, gcn_layers: int = 5, vae_hidden_dim: int = 512,
                 vae_latent_dim: int = 32, gat_hidden_channels: int = 64, property_embedding_dim: int = 8,
                 mlp_features=32, *args, **kwargs):
        super().__init__(vae_input_dim, device, gcn_layers, vae_hidden_dim, vae_latent_dim, gat_hidden_channels,
                         property_embedding_dim=property_embedding_dim, mlp_features=mlp_features)


class HybridModelv2(MultimodalNet):
    """reference ``hybrid_models.py:240-359``"""
    SPEC = Spec(attn="mha", comb=16)

    def __init__(self, vae_input_dim, device, gcn_layers: int = 5, vae_hidden_dim: int = 512,
                 vae_latent_dim: int = 32, gat_hidden_channels: int = 64, self_attention_heads: int = 1,
                 property_embedding_dim: int = 8, combined_attention_heads: int = 8, *args, **kwargs):
        super().__init__(vae_input_dim, device, gcn_layers, vae_hidden_dim, vae_latent_dim, gat_hidden_channels,
                         self_attention_heads, property_embedding_dim, combined_attention_heads)


class HybridModelv2_SSL(MultimodalNet):
    """reference ``hybrid_models.py:361-488``"""
    SPEC = Spec(attn="mha", comb=32, ssl=True)

    def __init__(self, vae_input_dim, device, gcn_layers: int = 5, vae_hidden_dim: int = 512,
                 vae_latent_dim: int = 32, gat_hidden_channels: int = 64, property_embedding_dim: int = 8,
                 self_attention_heads: int = 1, combined_attention_heads: int = 8, mlp_features: int = 32,
                 *args, **kwargs):
        super().__init__(vae_input_dim, device, gcn_layers, vae_hidden_dim, vae_latent_dim, gat_hidden_channels,
                         self_attention_heads, property_embedding_dim, combined_attention_heads,
                         mlp_features=mlp_features)
