"""Drop-in for the external package GammaGL's fused GAT layer imports: ``from dgNN.operators import GATConvFuse``
(gammagl/layers/conv/fusedgat_conv.py:70-71).  Put THIS directory on ``sys.path`` as ``dgNN`` (copy / symlink
``gammagl_amd/compat/dgNN`` next to the GammaGL checkout, or add ``gammagl_amd/compat`` to ``PYTHONPATH``) and
``FusedGATConv`` runs on the MI355X kernels with zero edits to GammaGL — see operators.py."""
