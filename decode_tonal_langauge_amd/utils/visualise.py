"""Plot helpers needed by the synthesis entry point.

``plot_training_losses`` is imported by the reference's ``train_synthesizer.py:21`` but does not
exist in its ``utils/visualise.py`` (the script raises ImportError as shipped, SURVEY.md section 0
finding 5); it is provided here, together with ``plot_confusion_matrix`` used by the classifier
pipeline.  The reference's other plotting helpers are out of scope."""
import os
from typing import List, Optional, Sequence

import numpy as np


def plot_training_losses(losses: Sequence[Sequence[float]], figure_path: Optional[str] = None,
                         labels: Optional[List[str]] = None):
    """One curve of per-epoch training loss per repeat; saves to ``figure_path`` if given."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    fig, ax = plt.subplots(figsize=(8, 5))
    for i, curve in enumerate(losses):
        ax.plot(range(1, len(curve) + 1), curve, label=(labels[i] if labels else f"run {i + 1}"))
    ax.set_xlabel("Epoch")
    ax.set_ylabel("Training loss (L1)")
    ax.set_title("Synthesizer training loss")
    if len(losses) <= 10:
        ax.legend()
    fig.tight_layout()
    if figure_path:
        fig.savefig(figure_path, dpi=150)
    plt.close(fig)
    return figure_path


def plot_confusion_matrix(confusion_matrix: np.ndarray, add_numbers: bool = False,
                          label_names: Optional[Sequence[str]] = None, figure_path: Optional[str] = None,
                          cmap: str = "Blues", title: str = "Confusion Matrix",
                          imshow_kwargs: Optional[dict] = None) -> None:
    """Heat map of a confusion matrix, rows = true label (reference utils/visualise.py:12-90)."""
    import matplotlib
    if figure_path is not None:
        matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    cm = np.asarray(confusion_matrix)
    fig, ax = plt.subplots(figsize=(8, 6))
    im = ax.imshow(cm, interpolation="nearest", cmap=cmap, **(imshow_kwargs or {}))
    ax.set_title(title, fontsize=20)
    fig.colorbar(im, ax=ax)
    ax.set_xlabel("Predicted Label", fontsize=18)
    ax.set_ylabel("True Label", fontsize=18)
    ticks_x, ticks_y = np.arange(cm.shape[1]), np.arange(cm.shape[0])
    if label_names is not None:
        ax.set_xticks(np.arange(len(label_names)), labels=list(label_names), rotation=45, fontsize=14)
        ax.set_yticks(np.arange(len(label_names)), labels=list(label_names), fontsize=14)
    else:
        ax.set_xticks(ticks_x)
        ax.set_yticks(ticks_y)
    if add_numbers:
        half = cm.max() / 2.0 if cm.size else 0.0
        for i in ticks_y:
            for j in ticks_x:
                v = cm[i, j]
                ax.text(j, i, f"{v:.0f}" if float(v).is_integer() else f"{v:.2f}", ha="center", va="center",
                        color="white" if v > half else "black")
    fig.tight_layout()
    if figure_path is not None:
        os.makedirs(os.path.dirname(figure_path) or ".", exist_ok=True)
        fig.savefig(figure_path)
        plt.close(fig)
    else:
        plt.show()
