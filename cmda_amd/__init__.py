"""cmda_amd -- MI355X (gfx950) implementation of CMDA's dense-segmentation training hot path.

Importing the package registers the modules under the reference's registry keys (cmda_amd.registry)."""
from . import registry  # noqa: F401
from . import backbones, decode_heads, fusion, segmentors, uda  # noqa: F401,E402
from . import datasets  # noqa: F401,E402  (registers UDADataset / CityscapesICDataset / DSECDataset / DarkZurichICDataset)
from .datasets import build_dataloader, build_dataset  # noqa: F401,E402
