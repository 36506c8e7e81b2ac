"""CPU oracle of the CLIP text encoder behind `pipe.encode_prompt` (dift_sd.py:258-263) - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product path never does.

Third-party arithmetic (not under /root/reference): HF transformers CLIPTextModel (modeling_clip.py CLIPTextTransformer:
token + position embeddings, causal-masked pre-LN encoder layers, final_layer_norm); the reference pins
transformers==4.31.0 / 4.38.2, the installed 5.x is the de-facto oracle.  Pinned against it with a tiny random-init
config: tests/golden/text_tiny.npz (tests/golden/make_golden.py gen_text).  Weight names = HF state-dict names without the
leading `text_model.`.
"""
import torch
import torch.nn.functional as F


def _act(x, kind):
    if kind == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)
    if kind == "gelu":
        return F.gelu(x)
    raise ValueError(kind)


def clip_text_hidden(w, input_ids, heads, act="quick_gelu", eps=1e-5, hidden_state=None):
    """input_ids [B, L] -> last_hidden_state [B, L, d] (after final_layer_norm); hidden_state=-2 -> HF
    `output_hidden_states=True).hidden_states[-2]` (output of the second-to-last layer, no final LN): what SDXL's
    encode_prompt takes from both of its text encoders (pipeline_stable_diffusion_xl.py encode_prompt)."""
    B, L = input_ids.shape
    h = w["embeddings.token_embedding.weight"][input_ids] + w["embeddings.position_embedding.weight"][:L][None]
    d = h.shape[-1]
    dh = d // heads
    mask = torch.full((L, L), float("-inf")).triu(1).to(h.dtype)
    n_layers = 1 + max(int(k.split(".")[2]) for k in w if k.startswith("encoder.layers."))
    if hidden_state is not None:
        n_layers = n_layers + 1 + hidden_state               # hidden_states[k] = input of layer k; [-2] skips the last layer
    for i in range(n_layers):
        p = f"encoder.layers.{i}"
        lin = lambda x, n: F.linear(x, w[f"{p}.{n}.weight"], w[f"{p}.{n}.bias"])
        n1 = F.layer_norm(h, (d,), w[f"{p}.layer_norm1.weight"], w[f"{p}.layer_norm1.bias"], eps)
        sp = lambda t: t.view(B, L, heads, dh).transpose(1, 2)
        q, k, v = sp(lin(n1, "self_attn.q_proj")), sp(lin(n1, "self_attn.k_proj")), sp(lin(n1, "self_attn.v_proj"))
        a = torch.softmax((q @ k.transpose(-1, -2)) * dh ** -0.5 + mask, dim=-1) @ v
        h = h + lin(a.transpose(1, 2).reshape(B, L, d), "self_attn.out_proj")
        n2 = F.layer_norm(h, (d,), w[f"{p}.layer_norm2.weight"], w[f"{p}.layer_norm2.bias"], eps)
        h = h + lin(_act(lin(n2, "mlp.fc1"), act), "mlp.fc2")
    if hidden_state is not None:
        return h
    return F.layer_norm(h, (d,), w["final_layer_norm.weight"], w["final_layer_norm.bias"], eps)
