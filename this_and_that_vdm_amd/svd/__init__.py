from .unet_spatio_temporal_condition import UNetSpatioTemporalConditionModel, UNetSpatioTemporalConditionOutput  # noqa: F401
from .temporal_controlnet import ControlNetModel, ControlNetOutput  # noqa: F401
from .scheduling_euler_discrete import EulerDiscreteScheduler  # noqa: F401
from .denoise import DenoiseLoop  # noqa: F401
from .pipeline_stable_video_diffusion import StableVideoDiffusionPipeline  # noqa: F401
from .pipeline_stable_video_diffusion_controlnet import StableVideoDiffusionControlNetPipeline  # noqa: F401
