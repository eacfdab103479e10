"""Cancer / wild-type paired models -- reference ``models/comparative_models.py``."""
from ._core import MultimodalNet, Spec

__all__ = ["HybridModel_Comparative", "HybridModelv2_Comparative", "HybridModelv2_Comparative_SSL",
           "HybridModel_Comparative_SSL"]


class HybridModel_Comparative(MultimodalNet):
    """reference ``comparative_models.py:11-173``"""
    SPEC = Spec(attn="v1", paired=True)

    def __init__(self, vae_input_dim, device, gcn_layers: int = 5, vae_hidden_dim: int = 512,
                 vae_latent_dim: int = 32, gat_hidden_channels: int = 64, property_embedding_dim: int = 8,
                 use_wt_for_downstream: bool = True):
        super().__init__(vae_input_dim, device, gcn_layers, vae_hidden_dim, vae_latent_dim, gat_hidden_channels,
                         property_embedding_dim=property_embedding_dim, use_wt_for_downstream=use_wt_for_downstream)


class HybridModel_Comparative_SSL(MultimodalNet):
    """reference ``comparative_models.py:175-350``"""
    SPEC = Spec(attn="v1", paired=True, ssl=True)

    def __init__(self, vae_input_dim, device, gcn_layers: int = 5, vae_hidden_dim: int = 512,
                 vae_latent_dim: int = 32, gat_hidden_channels: int = 64, property_embedding_dim: int = 8,
                 use_wt_for_downstream: bool = True, mlp_features: int = 32):
        super().__init__(vae_input_dim, device, gcn_layers, vae_hidden_dim, vae_latent_dim, gat_hidden_channels,
                         property_embedding_dim=property_embedding_dim, use_wt_for_downstream=use_wt_for_downstream,
                         mlp_features=mlp_features)


class HybridModelv2_Comparative(MultimodalNet):
    """reference ``comparative_models.py:353-527``"""
    SPEC = Spec(attn="mha", comb=32, paired=True)

    def __init__(self, vae_input_dim, device, gcn_layers: int = 5, vae_hidden_dim: int = 512,
                 vae_latent_dim: int = 32, gat_hidden_channels: int = 64, property_embedding_dim: int = 8,
                 self_attention_heads: int = 1, combined_attention_heads: int = 8,
                 use_wt_for_downstream: bool = True):
        super().__init__(vae_input_dim, device, gcn_layers, vae_hidden_dim, vae_latent_dim, gat_hidden_channels,
                         self_attention_heads, property_embedding_dim, combined_attention_heads,
                         use_wt_for_downstream)


class HybridModelv2_Comparative_SSL(MultimodalNet):
    """reference ``comparative_models.py:529-713``"""
    SPEC = Spec(attn="mha", comb=32, paired=True, ssl=True)

    def __init__(self, vae_input_dim, device, gcn_layers: int = 5, vae_hidden_dim: int = 512,
                 vae_latent_dim: int = 32, gat_hidden_channels: int = 64, property_embedding_dim: int = 8,
                 self_attention_heads: int = 1, combined_attention_heads: int = 8,
                 use_wt_for_downstream: bool = True, mlp_features: int = 32):
        super().__init__(vae_input_dim, device, gcn_layers, vae_hidden_dim, vae_latent_dim, gat_hidden_channels,
                         self_attention_heads, property_embedding_dim, combined_attention_heads,
                         use_wt_for_downstream, mlp_features)
