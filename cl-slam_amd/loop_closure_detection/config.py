"""Configuration section ``LoopClosureDetection`` of config_adapt.yaml:43-46, field for field the reference's
dataclass (loop_closure_detection/config.py:5-10) -- config/config_parser.py:40-91 coerces and fills it by name."""
import dataclasses
from pathlib import Path


@dataclasses.dataclass
class LoopClosureDetection:
    config_file: Path
    detection_threshold: float
    id_threshold: int
    num_matches: int
