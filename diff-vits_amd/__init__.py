"""MI355X-native diffusion-sampling path for diff-vits.

Public surface (mirrors the reference's import paths):
  diff_vits_amd.unet1d.unet_1d_condition.UNet1DConditionModel
  diff_vits_amd.unet1d.embeddings.TextTimeEmbedding
  diff_vits_amd.sampler.dpm_solver.{NoiseScheduleVP, model_wrapper, DPM_Solver}
  diff_vits_amd.sampler.uni_pc.{NoiseScheduleVP, model_wrapper, UniPC}
Putting this directory itself on sys.path makes `unet1d` / `sampler` importable under the
reference's own top-level names.
"""
__version__ = "0.1.0"
