"""One parametrised implementation behind the 14 reference model classes.

The reference spells every variant out as its own ``nn.Module`` with a copy of
the same forward body (``models/hybrid_models.py``, ``comparative_models.py``,
``ablation_models.py``).  Here a single ``MultimodalNet`` is configured by a
small spec (which encoders exist, which node attention, whether the fused
scalars go through the "combined attention", SSL heads, paired mode); the
public classes in the sibling modules only pin the spec and the constructor
signature.  Sub-module NAMES and shapes are the reference's, so
``state_dict()`` is key-for-key compatible.

Graph work runs on the HIP kernels (``immunostruct_amd.nn.EGNNConv`` and
``functional.segment_pool``); there is no CPU path.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as HF
from ..graph import PackedGraphBatch
from ..graph import batch as graph_batch
from ..nn import EGNNConv, egnn_stack_forward, egnn_stack_prelaunch, stack_is_native
from .layers import MultiHeadAttention, SelfAttention

NODE_ONEHOT = 20  # amino-acid one-hot columns of ndata['x'] (data/preprocess.py:40-41)
OVERLAP_BRANCHES = True      # the sequence branch on a forked stream (module constants, not switches: tests set them to compare forms)
EARLY_JOIN = True            # the fusion head joins the sequence branch at the latent (engine steps whose loss promised to, below)
JOIN_COUNTS = {"early": 0, "full": 0}      # how the main stream joined the sequence branch, per forward (tests)
# the sequence branch starts when this layer (0-based) of the EGNN stack has finished (clamped to the last layer).  "auto": its
# forward (~100 us of side work) should end with the stack + node attention, not stretch more layer launches than it must -- the
# longer a layer launch, the earlier the fork.  Measured on the round-3 kernels (same box, ms per step): B = 128 graphs / 72 k edges
# after layer 1 / 2 / 3 / 4 = 1.104 / 1.104 / 1.090 / 1.121; 128 pairs / 145 k edges after 2 / 3 = 2.123 / 2.143
_fork_env = os.environ.get("IMMUNOSTRUCT_FORK_AFTER_LAYER", "auto")
FORK_AFTER_LAYER = None if _fork_env == "auto" else int(_fork_env)
FORK_AUTO_EDGES = 100_000      # batches with at least this many edges fork one layer earlier


def fork_after_layer(num_edges):
    if FORK_AFTER_LAYER is not None:
        return FORK_AFTER_LAYER
    return 2 if num_edges >= FORK_AUTO_EDGES else 3
MERGE_PAIRS = True      # paired models: one encoder pass over [cancer; wild-type]
if OVERLAP_BRANCHES and hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
    # the sequence branch runs on a forked stream by design; autograd's per-call warning about it is noise here
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
_side_streams = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


class PairList(list):
    """[cancer, wild-type] as the reference returns it; ``merged`` = the tensor of the one encoder pass both are slices
    of (None after two passes).  The reconstruction / KLD terms of a pair, 0.5 * (term(cancer) + term(wild-type)) with
    equal member sizes, are the same means taken over ``merged`` -- one loss launch instead of two, and no slice
    backward (procedures.train._paired_loss)."""

    def __init__(self, items, merged=None):
        super().__init__(items)
        self.merged = merged


@dataclass(frozen=True)
class Spec:
    graph: bool = True          # EGNN + node attention + pooling branch
    attn: str = "mha"           # "v1" (SelfAttention) | "mha" (heads from ctor) | "mha8"
    vae: bool = True            # sequence VAE branch
    prop: str = "emb"           # "emb" (property MLP) | "raw" (concat raw 2-d) | "none"
    comb: int = 0               # feature width of the combined attention (0 = absent)
    ssl: bool = False           # classifier_head + node_predictor_head
    paired: bool = False        # cancer / wild-type comparative model
    pool: str = "mean"          # "mean" | "meanmax"


class MultimodalNet(nn.Module):
    SPEC = Spec()

    def __init__(self, vae_input_dim, device, gcn_layers=5, vae_hidden_dim=512, vae_latent_dim=32,
                 gat_hidden_channels=64, self_attention_heads=1, property_embedding_dim=8,
                 combined_attention_heads=8, use_wt_for_downstream=True, mlp_features=32):
        super().__init__()
        sp = self.SPEC
        self.device = device
        self.vae_input_dim = vae_input_dim
        self.vae_hidden_dim = vae_hidden_dim
        self.vae_latent_dim = vae_latent_dim
        self.gat_hidden_channels = gat_hidden_channels
        self.property_embedding_dim = property_embedding_dim
        self.use_wt_for_downstream = use_wt_for_downstream
        self.mlp_features = mlp_features
        self._pair_rows = 0       # > 0 while a merged (cancer; wild-type) batch is being encoded: rows of one member
        c = gat_hidden_channels

        if sp.graph:
            layers = [EGNNConv(NODE_ONEHOT, c, c, 1)]
            layers += [EGNNConv(c, c, c, 1) for _ in range(gcn_layers)]
            self.GCN_layers = nn.ModuleList(layers)
            if sp.attn == "v1":
                self.self_attention = SelfAttention(c)
            else:
                self.self_attention = MultiHeadAttention(c, 8 if sp.attn == "mha8" else self_attention_heads)
        if sp.vae:
            extra = {"emb": property_embedding_dim, "raw": 2, "none": 0}[sp.prop]
            self.vae_fc1 = nn.Linear(vae_input_dim, vae_hidden_dim)
            self.vae_fc21 = nn.Linear(vae_hidden_dim, vae_latent_dim)
            self.vae_fc22 = nn.Linear(vae_hidden_dim, vae_latent_dim)
            self.vae_fc3 = nn.Linear(vae_latent_dim + extra, vae_hidden_dim)
            self.vae_fc4 = nn.Linear(vae_hidden_dim, vae_input_dim)
        if sp.comb:
            self.combined_attention = MultiHeadAttention(sp.comb, combined_attention_heads, input_dim=1)
        self.classifier = self.get_classifier()
        if sp.ssl:
            self.classifier_head = nn.Linear(mlp_features, 1)
            self.node_predictor_head = nn.Linear(mlp_features, NODE_ONEHOT)
        if sp.prop == "emb":
            self.property_embedding = nn.Sequential(
                nn.Linear(2, 32), nn.ReLU(True), nn.Dropout(0.1),
                nn.Linear(32, property_embedding_dim), nn.ReLU(True))

    # ---- pieces with reference-visible names ------------------------------
    def _fused_width(self):
        sp = self.SPEC
        w = 0
        if sp.graph:
            w += self.gat_hidden_channels * (2 if sp.pool == "meanmax" else 1)
        if sp.vae:
            w += self.vae_latent_dim + {"emb": self.property_embedding_dim, "raw": 2, "none": 0}[sp.prop]
        if sp.paired and self.use_wt_for_downstream:
            w *= 2
        return w

    def get_classifier(self):
        mods = [nn.Flatten(1), nn.Linear(self._fused_width(), 32), nn.ReLU(True), nn.Dropout(0.1)]
        if not self.SPEC.ssl:
            mods.append(nn.Linear(32, 1))
        return nn.Sequential(*mods)

    def encode_vae(self, x):
        h1 = F.relu(HF.linear_small_batch(x, self.vae_fc1.weight, self.vae_fc1.bias))
        return self.vae_fc21(h1), self.vae_fc22(h1)

    def reparameterize(self, mu, logvar):
        # sampled on every call, also in eval mode (reference hybrid_models.py:301-304)
        if self._pair_rows:
            # merged (cancer; wild-type) batch: two draws in the reference's order (cancer first), one per member
            b = self._pair_rows
            eps = torch.cat([HF.randn_like(mu[:b]), HF.randn_like(mu[b:])], dim=0)
        else:
            eps = HF.randn_like(mu)
        return mu + eps * torch.exp(0.5 * logvar)

    def decode_vae(self, z):
        return HF.linear_small_batch(F.relu(self.vae_fc3(z)), self.vae_fc4.weight, self.vae_fc4.bias)

    def load_trained(self, path, new_head=False, map_location=None):
        self.load_state_dict(torch.load(path, map_location=map_location))
        if new_head:
            if self.SPEC.ssl:
                self.classifier_head = nn.Linear(self.mlp_features, 1).to(self.device)
            else:
                self.classifier = self.get_classifier().to(self.device)

    # ---- encoders ------------------------------------------------------------
    def _graph_inputs(self, g):
        feats = g.ndata["x"]
        layers = list(self.GCN_layers)
        head = self.self_attention.qk_head() if (self.SPEC.pool == "mean" and HF.fused_head_available(len(layers))) else None
        return feats[:, :NODE_ONEHOT], feats[:, NODE_ONEHOT:], g.edata["edge_attr"], layers, head

    def _encode_graph(self, g, need_attention=False, prologue=None):
        h, x, a, layers, head = prologue[0] if prologue is not None else self._graph_inputs(g)
        pro = prologue[1] if prologue is not None else None
        qk = None
        if head is not None:
            # the node attention's query / key projection rides on the last EGNN layer's node kernel
            h, x, qk = egnn_stack_forward(layers, g, h, x, a, head=head, final_coords=False, prologue=pro)
        else:
            # all layers, fused HIP kernels; the models keep only h (reference hybrid_models.py:323-324), so the last
            # layer's coordinate update is not evaluated
            h, x = egnn_stack_forward(layers, g, h, x, a, final_coords=False, prologue=pro)
        HF.StackBoundary.record(h, x, qk)
        c = self.gat_hidden_channels
        if g.uniform_nodes_per_graph() is None:
            raise ValueError("all graphs of a batch must be padded to the same node count "
                             "(reference data/preprocess.py:343-349)")
        hb = h.view(g.batch_size, -1, c)
        if self.SPEC.pool == "mean":
            # all graphs are padded to the same node count (checked above), so global_mean_pool over the
            # attention output is a plain mean over the n rows -- taken inside the attention block
            pooled, weights = self.self_attention.pooled_mean(hb, need_weights=need_attention, qk=qk)
        else:
            out, weights = self.self_attention(hb)
            pooled = HF.segment_pool(out.reshape(-1, c), g.seg_ptr(), self.SPEC.pool)
        return pooled, weights

    def _encode_sequence(self, seq, prop):
        """property MLP + sequence VAE (everything that does not depend on the graph)."""
        sp = self.SPEC
        o = {}
        p = None
        if sp.prop == "emb":
            p = HF.sequential_mlp2(self.property_embedding, prop)      # one HIP launch (csrc/mlp_head.hip)
            if p is None:
                p = self.property_embedding(prop)
        elif sp.prop == "raw":
            p = prop
        if sp.vae:
            x = seq.reshape(-1, self.vae_input_dim)
            fuse1 = (HF.vae_latent_supported(x, self.vae_latent_dim, p, hidden=self.vae_fc1.out_features)
                     and self.vae_fc21.bias is not None)
            a1 = None if fuse1 else HF.linear_small_batch(x, self.vae_fc1.weight, self.vae_fc1.bias)
            if fuse1 or (HF.vae_latent_supported(a1, self.vae_latent_dim, p) and self.vae_fc21.bias is not None):
                # fc21 | fc22, reparameterisation, cat(p), fc3 and the two ReLUs as ONE launch (csrc/vae_latent.hip); the noise
                # is drawn exactly as the reference does (torch.randn_like of a (B, latent) tensor; two draws for a merged pair).
                # With an input that needs no gradient (always, in the models) vae_fc1 runs inside the same autograd node, so
                # that the node's backward can order its launches: data path, vae_fc1's weight gradient, small weight pass.
                like = x.new_empty(x.shape[0], self.vae_latent_dim)
                if self._pair_rows:
                    b = self._pair_rows
                    eps = torch.cat([HF.randn_like(like[:b]), HF.randn_like(like[b:])], dim=0)
                else:
                    eps = HF.randn_like(like)
                mu, logvar, z, h3 = HF.vae_latent(a1, self.vae_fc21.weight, self.vae_fc21.bias, self.vae_fc22.weight,
                                                  self.vae_fc22.bias, eps, p, self.vae_fc3.weight, self.vae_fc3.bias,
                                                  fc1=(x, self.vae_fc1.weight, self.vae_fc1.bias) if fuse1 else None)
                self._latent_done(o, prop)
                recon = HF.linear_small_batch(h3, self.vae_fc4.weight, self.vae_fc4.bias)
            else:
                h1 = F.relu(a1)
                mu, logvar = self.vae_fc21(h1), self.vae_fc22(h1)
                z = self.reparameterize(mu, logvar)
                if p is not None:
                    z = torch.cat([z, p], dim=1)
                self._latent_done(o, prop)
                recon = self.decode_vae(z)
            o.update(mu=mu, logvar=logvar, z_vae=z, recon_x=recon)
        else:
            self._latent_done(o, prop)
        return o

    def _latent_done(self, o, prop):
        """everything the fusion head needs from the sequence branch exists (latent + property embedding): draw the classifier's
        dropout mask -- on the sequence branch's stream, long before the head needs it; the decoder draws nothing, so the
        order of the random draws is the reference's -- and mark the point: the head may start here, it does not need the
        reconstruction (``_encode``: ``EARLY_JOIN``)"""
        if not self.SPEC.ssl and prop.is_cuda:
            rows = prop.shape[0] // 2 if self._pair_rows else prop.shape[0]
            o["_cls_mask"] = (HF.sequential_dropout_mask(self.classifier, rows, prop.device),)
        if prop.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            o["_latent_ready"] = ev

    def _encode(self, g, seq, prop, need_attention=False):
        """Graph branch and sequence branch are independent until the fusion head: the sequence branch is
        enqueued on a side HIP stream so that its (library) GEMMs overlap the latency-bound graph kernels --
        also inside a captured HIP graph (fork/join) and, through autograd's stream bookkeeping, in backward."""
        sp = self.SPEC
        o = {}
        overlap = sp.graph and sp.vae and seq.is_cuda and OVERLAP_BRANCHES
        pro = None
        if overlap:
            # The EGNN stack's forward kernels are enqueued FIRST -- outside autograd; the stack's autograd node is created
            # further down, after the sequence branch's, so the backward still runs the graph branch's nodes first -- and the
            # sequence branch is forked from an event recorded part-way through the stack:
            #  * in a captured HIP graph the layer chain is then the first-captured continuation at every node, and ROCm's graph
            #    executor (first continuations stay on the launch queue, later ones get their own) runs the graph branch, the
            #    head, the loss and the main backward on ONE queue -- no cross-queue hop on the critical chain;
            #  * the sequence branch (and, behind it on the same stream, the speculative backward of its reconstruction term)
            #    runs beside the second half of the stack and the attention / head / loss kernels (whose small grids leave
            #    half the CUs idle), and its backward entirely between the head's and the stack's backward -- not beside the
            #    backward layer kernels, which own every wave slot (HISTORY.md section 3; the fork point is a measured sweep).
            main = torch.cuda.current_stream()
            side = _side_stream(seq.device)
            inp = self._graph_inputs(g)
            if stack_is_native(inp[3]):
                pro = (inp, egnn_stack_prelaunch(inp[3], g, inp[0], inp[1], inp[2], head=inp[4], final_coords=False,
                                                 fork_after=fork_after_layer(g.num_edges())))
                side.wait_event(pro[1].fork_event)
            else:
                side.wait_stream(main)      # layer sizes outside the HIP kernels' build: torch-composed stack, nothing to prelaunch
            with torch.cuda.stream(side):
                HF.Stamps.mark("fwd seq-branch start")
                o.update(self._encode_sequence(seq, prop))
                HF.Stamps.mark("fwd seq-branch end")
                HF.Stamps.hook(o["recon_x"], "bwd seq-branch start (d recon)")
        if sp.graph:
            HF.Stamps.mark("fwd graph-branch start")
            o["x_gat_node"], o["attention"] = self._encode_graph(g, need_attention, prologue=pro)
            HF.Stamps.mark("fwd graph-branch end")
            HF.Stamps.hook(o["x_gat_node"], "bwd graph-branch start (d x_gat)")
        if overlap:
            # the head needs the latent, not the reconstruction: when the loss consumes the reconstruction on the sequence branch's
            # own stream (functional.SpeculativeBackward: the engine's steps) the main stream joins at the latent -- an event long
            # passed when the node attention ends -- instead of behind the decoder's GEMM (a cross-queue wait of ~15 us in the
            # replayed step's timeline, round 3)
            ev = o.pop("_latent_ready", None)
            if (EARLY_JOIN and HF.SpeculativeBackward.enabled and HF.SpeculativeBackward.early_join and ev is not None
                    and torch.is_grad_enabled()):
                # (early_join: the step's loss promised to read the reconstruction through functional.vae_loss -- which runs on the
                #  branch's stream when it speculates, and waits for ``_ready_event`` when it does not)
                main.wait_event(ev)
                done = torch.cuda.Event()
                done.record(side)
                if torch.is_tensor(o.get("recon_x")):
                    o["recon_x"]._ready_event = done
                JOIN_COUNTS["early"] += 1
            else:
                main.wait_stream(side)
                JOIN_COUNTS["full"] += 1
            for t in o.values():
                for u in (t if isinstance(t, tuple) else (t,)):
                    if torch.is_tensor(u):
                        u.record_stream(main)
        else:
            o.update(self._encode_sequence(seq, prop))
            o.pop("_latent_ready", None)
        return o

    def _head(self, pieces, cls_mask=None):
        """``pieces``: the fused row as a list of (B, w) tensors laid side by side"""
        if self.SPEC.comb and not self.SPEC.ssl and pieces[0].is_cuda:
            # combined attention + classifier as ONE launch in each direction (csrc/combined_attention.hip)
            hid = HF.combined_attention_classifier(pieces, self.combined_attention, self.classifier,
                                                   mask=cls_mask[0] if cls_mask else "draw")
            if hid is not None:
                return hid, None
        if self.SPEC.comb:
            ca = self.combined_attention
            if ca.n_head == 8 and ca.w_q.in_features == 1 and sum(p.shape[1] for p in pieces) <= 256 and len(pieces) <= 4:
                # closed-form HIP kernel; reads the pieces where they are (no concatenation, no slice copies in backward)
                fused = HF.combined_attention_mean(pieces, ca)
            else:
                fused = ca(torch.cat(pieces, dim=1).unsqueeze(2))[0].mean(dim=2)
        else:
            fused = torch.cat(pieces, dim=1) if len(pieces) > 1 else pieces[0]
        hid = HF.sequential_mlp2(self.classifier, fused.flatten(1), mask=cls_mask[0] if cls_mask else "draw") if not self.SPEC.ssl else None
        if hid is None:
            hid = self.classifier(fused)
        if self.SPEC.ssl:
            return self.classifier_head(hid), self.node_predictor_head(hid)
        return hid, None

    def _pack(self, first, o, final, node_pred):
        if self.SPEC.vae:
            head = (first, o["mu"], o["logvar"], final)
        else:
            head = (0, 0, 0, final)
        return head + ((node_pred,) if self.SPEC.ssl else ())

    # ---- public forward passes --------------------------------------------
    def forward(self, graph_data, sequence_data, peptide_property, return_embedding=False, return_attention=False):
        sp = self.SPEC
        o = self._encode(graph_data, sequence_data, peptide_property, need_attention=return_attention)
        if sp.graph and sp.vae:
            parts = [o["x_gat_node"], o["z_vae"]]
            if sp.paired and self.use_wt_for_downstream:
                parts = parts * 2   # single-sample pretraining of a paired model
        else:
            parts = [o["x_gat_node"] if sp.graph else o["z_vae"]]
        final, node_pred = self._head(parts, o.get("_cls_mask"))
        first = o.get("recon_x")
        if sp.graph and sp.vae:
            if return_embedding:
                first = o["x_gat_node"]
            elif return_attention:
                first = o["attention"]
        return self._pack(first, o, final, node_pred)

    def forward_item(self, graph_data, sequence_data, peptide_property):
        o = self._encode(graph_data, sequence_data, peptide_property, need_attention=True)
        return o["mu"], o["logvar"], o["x_gat_node"], o["z_vae"], o["attention"], o["recon_x"]

    def _encode_pair(self, graphs, seqs, props, need_attention):
        """Encoder outputs of the cancer and of the wild-type member.  The encoder treats every graph / sample on its
        own (block-diagonal batch, per-graph attention, per-sample VAE and property MLP) and both members share its
        weights, so the pair is encoded as ONE batch of 2B graphs [cancer; wild-type] -- half the launches, and no
        gradient-accumulation kernels for the twice-used parameters -- and the outputs are split afterwards.
        Inputs: the reference's 2-tuples (merged here when both graphs are plain batches with the same node layout), or
        an already merged batch (one graph of 2B graphs, sequences / properties with 2B rows: what the on-GPU batcher
        delivers).  Static (capacity-padded) buffers take two encoder passes."""
        merged = None
        if isinstance(graphs, PackedGraphBatch):
            if graphs.batch_size % 2 or seqs.shape[0] != graphs.batch_size:
                raise ValueError("a merged pair batch holds 2B graphs and 2B sequence / property rows")
            merged = (graphs, seqs, props)
        elif (MERGE_PAIRS and self.SPEC.graph and self.SPEC.vae and type(graphs[0]) is PackedGraphBatch
              and type(graphs[1]) is PackedGraphBatch and graphs[0].batch_size == graphs[1].batch_size
              and graphs[0].num_nodes() == graphs[1].num_nodes() and seqs[0].shape == seqs[1].shape):
            graphs[0].csr(), graphs[1].csr()
            merged = (graph_batch([graphs[0], graphs[1]]), torch.cat([seqs[0], seqs[1]], dim=0), torch.cat([props[0], props[1]], dim=0))
        if merged is None:
            oc = self._encode(graphs[0], seqs[0], props[0], need_attention=need_attention)
            ow = self._encode(graphs[1], seqs[1], props[1])
            return oc, ow
        b = merged[0].batch_size // 2
        self._pair_rows = b
        try:
            o = self._encode(*merged, need_attention=need_attention)
        finally:
            self._pair_rows = 0
        halves = ({"_merged": o}, {"_merged": o})
        for k, v in o.items():
            for i, h in enumerate(halves):
                h[k] = v[i * b:(i + 1) * b] if torch.is_tensor(v) else v
        return halves

    def forward_comparative(self, graph_data_pair, sequence_data_pair, peptide_property_pair,
                            return_embedding=False, return_attention=False):
        if not self.SPEC.paired:
            raise AttributeError(f"{type(self).__name__} has no comparative forward")
        oc, ow = self._encode_pair(graph_data_pair, sequence_data_pair, peptide_property_pair, return_attention)
        final = None
        m = oc.get("_merged")
        if (m is not None and self.use_wt_for_downstream and self.SPEC.comb and not self.SPEC.ssl and m["x_gat_node"].is_cuda):
            # one encoder pass produced the stacked pair: the embeddings and the fusion head read the stacked tensors where the
            # rows are, and their backward returns ONE gradient per stacked tensor (no slice-backward / accumulate launches)
            xg, zv = m["x_gat_node"], m["z_vae"]
            b = xg.shape[0] // 2
            cls_mask = oc.get("_cls_mask")
            # (the embeddings FIRST: autograd runs later-created nodes earlier, so the head's backward is enqueued on the main
            #  stream in front of the embeddings' backward, which has to wait for the contrastive loss' backward on the side stream)
            emb_c, emb_w = HF.pair_embeddings(xg, zv, b)
            # marks the point of the stream at which the embeddings exist: a consumer on another stream (the contrastive loss,
            # procedures.train._contrastive_beside) waits for THIS, not for the head that is enqueued behind it
            emb_c._ready_event = torch.cuda.Event()
            emb_c._ready_event.record()
            final = HF.combined_attention_classifier([xg, zv], self.combined_attention, self.classifier,
                                                     mask=cls_mask[0] if cls_mask else "draw", pair_rows=b)
            node_pred = None
        if final is None:
            emb_c = torch.cat([oc["x_gat_node"], oc["z_vae"]], dim=1)      # returned to the caller (contrastive loss)
            emb_w = torch.cat([ow["x_gat_node"], ow["z_vae"]], dim=1)
            pieces = [oc["x_gat_node"], oc["z_vae"]]
            if self.use_wt_for_downstream:
                pieces += [ow["x_gat_node"], ow["z_vae"]]
            final, node_pred = self._head(pieces, oc.get("_cls_mask"))
        tail = (node_pred,) if self.SPEC.ssl else ()
        if return_embedding:
            return (oc["x_gat_node"], oc["mu"], oc["logvar"], final) + tail
        if return_attention:
            return (oc["attention"], oc["mu"], oc["logvar"], final) + tail
        pair = lambda k: PairList([oc[k], ow[k]], oc.get("_merged", {}).get(k))
        return ([emb_c, emb_w], pair("recon_x"), pair("mu"), pair("logvar"), final) + tail
