"""SynthesisLite on MI355X - filled in below (csrc/tonal_lite.hip)."""


def lite_apply(model, x_ecog, x_label):
    raise NotImplementedError("SynthesisLite HIP kernels are not built yet")
