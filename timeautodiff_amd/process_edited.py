"""Device-side ``convert_to_tensor`` (reference process_edited.py:252-301): the step every driver runs on
the decoder's outputs right after sampling (``gen = ae.decoder(samples, mask, cond)``;
``convert_to_tensor(parser, gen, 1, B, T)``, e.g. Run_Imputation.ipynb:171-172).  Same signature and
result; the work is one launch of ``tad_heads_to_tensor`` instead of a Python loop over columns, and the
tensor stays on the device the decoder outputs live on (the reference hard-codes 'cuda').

``DeapStack.decode_to_tensor`` (VAE.py of this package) goes one step further and feeds the decoder's
fused head buffer to the kernel directly, without materialising the dict."""
import torch

from . import ops


def convert_to_tensor(parser, gen_output, threshold, data_size, seq_len):
    """-> [data_size, seq_len, n_bins + n_cats + n_nums] float32: bins thresholded at sigmoid > 0.5, categorical
    codes = argmax of the softmax, numericals unchanged.  ``threshold`` is accepted and ignored, as in the
    reference."""
    info = parser.datatype_info()
    nb, nc, nn, cards = info["n_bins"], info["n_cats"], info["n_nums"], list(info["cards"])
    nc = len(cards)                    # the reference loops over len(cards) (process_edited.py:290-293)
    segs = []
    if nb:
        segs.append(gen_output["bins"].detach())
    for j in range(nc):
        segs.append(gen_output["cats"][j].detach())
    if nn:
        segs.append(gen_output["nums"].detach())
    if not segs:
        raise ValueError("gen_output holds no columns")
    if data_size is None or seq_len is None:
        data_size, seq_len = segs[0].shape[0], segs[0].shape[1]
    if not segs[0].is_cuda:
        raise ops._lib.TadError("convert_to_tensor needs the decoder outputs on the GPU (no CPU path exists)")
    heads = torch.cat([s.to(torch.float32).reshape(data_size * seq_len, -1) for s in segs], dim=1)
    out = torch.empty(data_size, seq_len, nb + nc + nn, dtype=torch.float32, device=heads.device)
    ops.heads_to_tensor(heads, nb, nc, nn, cards, out.view(data_size * seq_len, -1))
    return out
