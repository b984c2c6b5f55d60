"""rustfst_amd — MI355X-native compose + shortest-path engine behind rustfst's interface.

Scope: `rustfst::algorithms::compose` and `rustfst::algorithms::shortest_path` for
VectorFst<TropicalWeight>.  The product is the C-ABI library
rustfst_amd/lib/libwfst_amd.so (include/wfst.h, hand-written HIP for gfx950); this package is
the thin Python mirror of the reference's Python surface used by tests and bench.py.
"""
from ._lib import LIB_PATH, TR_DTYPE, WfstError  # noqa: F401
from .fst import (  # noqa: F401
    ComposeConfig,
    ComposeFilter,
    Context,
    DeviceFst,
    ShortestPathConfig,
    Tr,
    VectorFst,
    acceptor,
    compose,
    compose_shortest_path_batch,
    compose_shortest_path_batch_begin,
    compose_shortest_path_batch_packed,
    shortest_path_batch,
    last_nbest_path,
    HandleArray,
    LookAhead,
    ProjectType,
    project,
    compose_with_config,
    default_context,
    set_default_context,
    shortestpath,
    shortestpath_with_config,
)

__all__ = [
    "ComposeConfig", "ComposeFilter", "Context", "DeviceFst", "ShortestPathConfig", "Tr", "VectorFst", "acceptor",
    "compose", "compose_shortest_path_batch", "compose_shortest_path_batch_begin", "compose_shortest_path_batch_packed", "shortest_path_batch", "last_nbest_path", "HandleArray", "LookAhead", "ProjectType", "project", "compose_with_config", "default_context", "set_default_context",
    "shortestpath", "shortestpath_with_config", "WfstError", "TR_DTYPE", "LIB_PATH",
]
