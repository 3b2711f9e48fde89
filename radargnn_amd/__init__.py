"""radargnn_amd -- MI355X-native hot path of RadarGNN: graph construction, feature extraction and the
DetNetBasic forward pass as hand-written gfx950 HIP kernels behind a C ABI (include/rgnn.h).

Sub-modules
    radargnn_amd.ops                 torch-tensor front end of the C ABI (device memory + streams only)
    radargnn_amd.graph_constructor   mirror of gnnradarobjectdetection.graph_constructor (Graph / GeometricGraph)
    radargnn_amd.gnn                 mirror of gnnradarobjectdetection.gnn (MPNNConv, RadarPointGNNConv, DetNetBasic)
    radargnn_amd.frames              batched on-device pipeline: frames in HBM -> graphs -> logits / boxes
    radargnn_amd.synthetic           deterministic synthetic radar frames (no dataset travels with the repo)
    radargnn_amd.checkpoint          reads the reference trainer's whole-module pickles without torch_geometric

There is no CPU fallback: importing ``radargnn_amd.ops`` without the built ``librgnn.so`` raises.
"""
__version__ = "0.1.0"


def load_reference_model(path, map_location="cpu"):
    """``trained_model.pt`` of the reference's trainer -> the HIP ``DetNetBasic`` (radargnn_amd.checkpoint)."""
    from .checkpoint import load_reference_model as _load
    return _load(path, map_location)
