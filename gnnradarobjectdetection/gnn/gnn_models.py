from radargnn_amd.gnn.gnn_models import DetNetBasic, get_mlp  # noqa: F401
from radargnn_amd.gnn.linear import BatchNorm, Linear  # noqa: F401
