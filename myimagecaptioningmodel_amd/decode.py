"""Host-side caption post-processing of the decode path (the reference's evaluate.py:14-25,30-33 and infer.py:34-36).

The decode output is a float32 id matrix [B, Ti] (quirk Q2) with no early stop (Q5): a caption is the ids up to, not
including, the first <stop>, with <pad> ids skipped; ids are rounded to integers first (evaluate.py:31-32)."""
import numpy as np


def ids_to_tokens(ids, stop_idx=3, pad_idx=0, index_word=None):
    """One row of decode ids -> list of token ids (or of words when `index_word` maps id -> word)."""
    out = []
    for v in np.rint(np.asarray(ids)).astype(np.int64).tolist():
        if v == stop_idx:
            break
        if v == pad_idx:
            continue
        out.append(index_word[v] if index_word is not None else v)
    return out


def words2sentence(words):
    """evaluate.py:40-41."""
    return ' '.join(str(w) for w in words)
