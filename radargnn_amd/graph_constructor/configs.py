"""Graph-construction settings, field-compatible with ``preprocessor/configs.py:4-26`` of the reference
(``GraphConstructionConfiguration``): positional construction as in test/test_preprocessor.py:219-220 works."""
from dataclasses import dataclass
from typing import List, Optional


@dataclass
class GraphConstructionConfiguration:
    graph_construction_algorithm: str          # "knn" | "radius"
    graph_construction_settings: dict          # {"k": int, "r": float}
    node_features: List[str]
    edge_features: List[str]
    edge_mode: str                             # "directed" | "undirected"
    distance_definition: str                   # "X" | "XV"

    def __post_init__(self):
        algo = self.graph_construction_algorithm
        if algo not in ("knn", "radius"):
            raise Exception("Invalid graph construction algorithm selected")
        self.k: Optional[int] = self.graph_construction_settings.get("k") if algo == "knn" else None
        self.r: Optional[float] = self.graph_construction_settings.get("r") if algo == "radius" else None
