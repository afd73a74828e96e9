"""Files a GammaGL checkout takes as they are: `_torch_ext.py` stands in for the pybind module
`gammagl/mpops/torch_ext/_torch_ext` so that `gammagl/mpops/torch.py:3-7` binds the MI355X backend with zero edits."""
