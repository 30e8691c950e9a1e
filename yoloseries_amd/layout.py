"""Cell-major ("NHWC") head-tensor layout shared by the loss and the evaluators.

The HIP models return head outputs as logically (B, C, h, w) tensors whose memory is
[B][h][w][ld] (ld = C padded to a multiple of 8): strides (h*w*ld, 1, w*ld, ld).  Kernels
consume such views in place; any other tensor is copied into this layout first."""
import torch


def is_cell_major(p):
    if p.dim() != 4:
        return False
    B, Ct, h, w = p.shape
    st = p.stride()
    return (p.dtype in (torch.bfloat16, torch.float32) and st[1] == 1 and st[3] % 8 == 0 and st[3] >= Ct
            and st[2] == w * st[3] and st[0] == h * w * st[3] and p.data_ptr() % 16 == 0)


def to_cell_major(p):
    """(B, C, h, w) any strides -> (view in cell-major layout, ld)."""
    if is_cell_major(p):
        return p, p.stride(3)
    B, Ct, h, w = p.shape
    ld = ((Ct + 7) // 8) * 8
    dt = p.dtype if p.dtype in (torch.bfloat16, torch.float32) else torch.float32
    buf = torch.zeros(B, h, w, ld, dtype=dt, device=p.device)
    buf[..., :Ct] = p.detach().permute(0, 2, 3, 1)
    return buf.as_strided((B, Ct, h, w), (h * w * ld, 1, w * ld, ld)), ld


def cell_major_view(buf, C):
    """[B][h][w][ld] buffer -> logical (B, C, h, w) view."""
    B, h, w, ld = buf.shape
    return buf.as_strided((B, C, h, w), (h * w * ld, 1, w * ld, ld))
