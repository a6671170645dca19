"""Model constants for the per-fragment 3D path.

Plain-Python stand-in for the yacs node the reference reads (config/default.py:3-75,
config/test.yaml:22-46).  Several behaviour switches are hard-coded constants in the reference
rather than config keys (models/neucon_network.py:37-38,60-71,240-244,463,550); they are named
here so every module reads them from one place.
"""
from dataclasses import dataclass, field
from typing import List


@dataclass
class FusionCfg:
    FUSION_ON: bool = True          # config/test.yaml:36
    HIDDEN_DIM: int = 64
    AVERAGE: bool = False
    FULL: bool = True               # config/test.yaml:40


@dataclass
class ModelCfg:
    N_LAYER: int = 3                                    # config/test.yaml:26
    N_VOX: List[int] = field(default_factory=lambda: [96, 96, 96])
    VOXEL_SIZE: float = 0.04
    TRAIN_NUM_SAMPLE: List[int] = field(default_factory=lambda: [15000, 60000, 120000])
    TEST_NUM_SAMPLE: List[int] = field(default_factory=lambda: [15000, 60000, 120000])
    THRESHOLDS: List[float] = field(default_factory=lambda: [0, 0, 0])
    POS_WEIGHT: float = 1.5
    ALPHA: int = 1                                      # BACKBONE2D.ARC 'fpn-mnas-1'
    FUSION: FusionCfg = field(default_factory=FusionCfg)
    SPARSEREG_DROPOUT: bool = False                     # config/default.py:69


# constants the reference hard-codes (file:line in the module docstring)
N_VIEWS = 9
INIT_STAGE = 1                 # occupancy initialisation runs on the 48^3 grid
INIT_MIN_VIEW = 2
INIT_OCC_THRESHOLD = 0.3       # sigmoid(logit) > 0.3
INIT_MIN_VALID = 10 * 10 * 10  # models/occupancy_initialization.py:107
CH_IMG = [80, 40, 24]          # backbone pyramid channels, coarse -> fine
CH_VOXEL = [96, 48, 24]        # SPVCNN output channels, coarse -> fine
CH_INIT_DOWN = 32
PANOPTIC_CH = 48
NUM_CLASSES = 20
NUM_QUERIES = 80
EXCEED_NUM = 1.5
STAGE_MIN_OCC = 500            # models/neucon_network.py:469
PANOPTIC_SHAPE = (96, 96, 96)
