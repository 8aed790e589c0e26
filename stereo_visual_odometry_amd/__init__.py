"""MI355X-native stereo visual-odometry front end (drop-in for Alex7Li/stereo_visual_odometry's hot path).

Importing the package does not touch the GPU; importing `stereo_visual_odometry_amd.api` loads
libsvo_hip.so and fails loudly if it has not been built.
"""
__all__ = ["synthetic"]
