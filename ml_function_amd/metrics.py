"""Evaluation metrics of the reference's training scripts: binary cross-entropy is torch's; AUC
(example/ctr_example/un_seq.py:61 compiles the model with tf.keras.metrics.AUC) is computed exactly here, as the
Mann-Whitney statistic with average ranks for ties -- the quantity sklearn.metrics.roc_auc_score returns (Keras' AUC is
a 200-threshold approximation of the same number)."""
import torch


def auc(y_true, y_score):
    """y_true [N] in {0,1}, y_score [N] -> ROC AUC (float).  Runs on the tensors' device."""
    y_true = y_true.reshape(-1).to(torch.float64)
    y_score = y_score.reshape(-1).to(torch.float64)
    n_pos = float(y_true.sum())
    n_neg = float(y_true.numel()) - n_pos
    if n_pos == 0 or n_neg == 0:
        raise ValueError("auc: only one class present")
    order = torch.argsort(y_score)
    s = y_score[order]
    # average 1-based ranks over runs of equal scores
    uniq, inverse, counts = torch.unique_consecutive(s, return_inverse=True, return_counts=True)
    ends = torch.cumsum(counts, 0).to(torch.float64)
    avg_rank = ends - (counts.to(torch.float64) - 1.0) / 2.0
    ranks = avg_rank[inverse]
    pos_rank_sum = float((ranks * y_true[order]).sum())
    return (pos_rank_sum - n_pos * (n_pos + 1.0) / 2.0) / (n_pos * n_neg)
