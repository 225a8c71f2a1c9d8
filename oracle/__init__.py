"""CPU oracle for the VQ-VAE voice-conversion training step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the timed CPU baseline.
The product (``crank_amd``) never imports this package and fails loudly when
its HIP library is missing.

It is a plain PyTorch-fp32 (CPU) restatement of

* the third-party ``parallel_wavegan`` networks crank builds its conv stacks
  from (un-vendored, un-pinned pip dependency; restated from its public
  definition)                      -> ``oracle/pwg.py``
* crank's own arithmetic (``crank/net/module/*.py``)      -> ``oracle/modules.py``

Pinning status
--------------
* ``oracle/modules.py`` (Quantizer incl. EMA, losses incl. the STFT argument
  quirk, STFT layer, scaler layer, gradient reversal, VQVAE2 wiring) is pinned
  against outputs of the reference's own classes imported in the authoring
  container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
* ``oracle/pwg.py`` follows the published definition of ``parallel_wavegan``;
  the package is absent from /root/reference and from this image, the
  reference holds no test that pins its outputs, so **conv-stack parity is
  unpinned** against the third-party code itself.  Whole-step golden vectors
  are produced by running the *reference's own* ``VQVAE2`` /
  ``SpeakerAdversarialNetwork`` / trainer classes with ``oracle/pwg.py``
  registered in place of the absent package.
* ``oracle/dataset.py`` (numpy: batch assembly, sklearn scalers, F0 conversion,
  decode-side features) is pinned bit-exact by ``tests/golden/dataset.npz``
  (the reference's own BaseDataset methods, convert_f0,
  BaseTrainer._store_features / _get_cvf0, sklearn's StandardScaler, scaler.pkl).
* ``oracle/mcd.py`` restates the third-party ``fastdtw`` package (absent,
  un-pinned, no vectors in the reference): **parity unpinned**; checked against
  exhaustive DTW only.
"""
