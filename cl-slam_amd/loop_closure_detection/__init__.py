"""MI355X-native ``loop_closure_detection`` package: the reference's exports (loop_closure_detection/__init__.py:1-3
-- ``utils``, ``Config``, ``LoopClosureDetection``) plus ``FeatureEncoder``, so that config/config_parser.py:11 and
slam/slam.py:10 import it unchanged when cl-slam_amd/ precedes the reference on sys.path.  The encoder forward
 and the cosine search run on the HIP kernels; faiss is not needed."""
import loop_closure_detection.utils
from loop_closure_detection.config import LoopClosureDetection as Config
from loop_closure_detection.encoder import FeatureEncoder
from loop_closure_detection.loop_closure_detection import LoopClosureDetection

__all__ = ['Config', 'FeatureEncoder', 'LoopClosureDetection', 'utils']
