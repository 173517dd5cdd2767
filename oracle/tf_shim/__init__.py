"""Stand-in `tensorflow` / `tensorflow_addons` / `gin` packages  --  TEST INFRASTRUCTURE ONLY.

Purpose: run the REFERENCE'S OWN Python (``/root/reference/models/film_net/*.py``,
``eval/interpolator.py``, ``eval/util.py``) in a container that has no TensorFlow, so that the graph
wiring, channel orders, pyramid bookkeeping, padding / patch code and the recursion order of the
golden vectors under ``tests/golden/ref_*.npz`` come from the reference itself and not from a
restatement.  Only the ~25 TF / TFA / Keras entry points those files call are provided, each
implemented with an independent PyTorch-CPU built-in:

    Conv2D 'same'/'valid'         F.conv2d (+ F.pad bottom/right for even kernels)
    AveragePooling2D(2,2,'valid') F.avg_pool2d
    tf.image.resize bilinear      F.interpolate(mode='bilinear', align_corners=False)
    tf.image.resize NEAREST       F.interpolate(mode='nearest')
    tfa.image.dense_image_warp    F.grid_sample(bilinear, padding_mode='border', align_corners=True)
    space_to_batch / batch_to_space / split / stack / pad_to_bounding_box / crop_to_bounding_box
                                  written from the TF API documentation (reshape / transpose / pad)

What this pins and what it does not: the GRAPH is the reference's code, executed; the OP SEMANTICS are
still a statement about TF 2.6.2 / TFA 0.15.0 (not installable here) - made a second time, by a
different route than oracle/film_oracle.py (library built-ins instead of hand-written index
arithmetic).  ``tools/make_ref_golden.py --backend tf`` runs the same script on real TensorFlow where
one exists.

Use: ``oracle.tf_shim.install()`` puts this directory at the front of ``sys.path``; nothing in the
product path (frame-interpolation_amd/) may import it.
"""
import os
import sys

SHIM_DIR = os.path.dirname(os.path.abspath(__file__))


def install() -> None:
    """Makes `import tensorflow`, `import tensorflow_addons.image`, `import gin.tf` resolve here."""
    for m in ('tensorflow', 'tensorflow_addons', 'gin'):
        if m in sys.modules and not getattr(sys.modules[m], '__film_shim__', False):
            raise RuntimeError(f'a real {m} is already imported; the shim is not needed')
    if SHIM_DIR not in sys.path:
        sys.path.insert(0, SHIM_DIR)
