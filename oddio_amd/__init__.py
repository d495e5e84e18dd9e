"""oddio_amd -- MI355X-native batched implementation of oddio's SpatialScene/Mixer hot path."""

from .api import (  # noqa: F401
    Adapt, AdaptOptions, Constant, Cycle, Downmix, Fader, FaderControl, FixedGain, Frames, FramesSignal, Gain, GainControl, Mixed, Mixer, MixerControl, MonoToStereo, Reinhard, Signal, Sine,
    Spatial, SpatialOptions, SpatialScene, SpatialSceneControl, Speed, SpeedControl, Stream, StreamControl, Tanh, frame_stereo, run,
    FILTER_FIXED_GAIN, FILTER_GAIN, FILTER_SPEED, MODE_FAST, MODE_FAST_UNFUSED, MODE_ORDERED, MODE_TRACKED, POSTFX_NONE, POSTFX_REINHARD, POSTFX_TANH,
)
