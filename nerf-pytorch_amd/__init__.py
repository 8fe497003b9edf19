"""nerf-pytorch_amd: the MI355X-native NeRF render + training hot path behind the krrish94/nerf-pytorch API."""
