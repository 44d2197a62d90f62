"""Plot helpers needed by the synthesis entry point.

``plot_training_losses`` is imported by the reference's ``train_synthesizer.py:21`` but does not
exist in its ``utils/visualise.py`` (the script raises ImportError as shipped, SURVEY.md section 0
finding 5); it is provided here.  The reference's other plotting helpers are out of scope."""
from typing import List, Optional, Sequence


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
