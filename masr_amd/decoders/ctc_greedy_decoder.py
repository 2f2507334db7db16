"""CTC greedy decoders with the reference's signatures (masr/decoders/ctc_greedy_decoder.py).
argmax / max-prob and the best-path collapse + score run in HIP kernels (masr_argmax_rows,
masr_ctc_collapse); this module only maps token ids to text."""
import numpy as np
import torch

from masr_amd import runtime


def _frames(probs_seq):
    """np probs [T, V] -> per-frame (argmax ids, max probs) computed on the GPU."""
    eng = runtime.aux_engine()
    p = torch.from_numpy(np.ascontiguousarray(np.asarray(probs_seq), dtype=np.float32)).to(eng.device)
    if p.shape[0] == 0:
        return eng, torch.zeros(0, dtype=torch.int32, device=eng.device), torch.zeros(0, device=eng.device)
    idx, mp = eng.argmax_rows(p)
    return eng, idx, mp


def _collapse(eng, idx, mp, vocabulary, blank_index):
    T = idx.shape[0]
    if T == 0:
        return 0, ''
    tok, ntok, score = eng.ctc_collapse(idx.view(1, T), mp.view(1, T), None, blank_index)
    n = int(ntok[0])
    ids = tok[0, :n].cpu().numpy()
    text = ''.join([vocabulary[i] for i in ids])
    has_non_blank = bool((idx != blank_index).any())
    sc = float(np.float32(score[0].item())) * 100.0 if has_non_blank else 0
    return sc, text.replace('<space>', ' ')


def greedy_decoder(probs_seq, vocabulary, blank_index=0):
    """ctc_greedy_decoder.py:6-31 -> (score, text)."""
    eng, idx, mp = _frames(probs_seq)
    return _collapse(eng, idx, mp, vocabulary, blank_index)


def greedy_decoder_batch(probs_split, vocabulary, blank_index=0):
    """ctc_greedy_decoder.py:34-49 -> list of texts."""
    return [greedy_decoder(p, vocabulary, blank_index)[1] for p in probs_split]


def greedy_decoder_chunk(probs_seq, vocabulary, last_max_prob_list=None, last_max_index_list=None, blank_index=0):
    """ctc_greedy_decoder.py:52-89.  State lists keep the reference's (swapped) naming:
    ``last_max_prob_list`` holds all frame INDICES so far, ``last_max_index_list`` the max
    PROBS of the non-blank frames; text and score are recomputed over the full history."""
    if last_max_prob_list is None:
        last_max_prob_list = []
    if last_max_index_list is None:
        last_max_index_list = []
    eng, idx, mp = _frames(probs_seq)
    ids = idx.cpu().numpy()
    mps = mp.cpu().numpy()
    last_max_prob_list.extend(int(i) for i in ids)
    last_max_index_list.extend(np.float32(p) for i, p in zip(ids, mps) if i != blank_index)
    return decode_history(last_max_prob_list, last_max_index_list, vocabulary, blank_index) + \
        (last_max_prob_list, last_max_index_list)


def greedy_decoder_chunk_frames(ids, max_probs, vocabulary, last_max_prob_list=None, last_max_index_list=None, blank_index=0):
    """greedy_decoder_chunk for a chunk whose per-frame (argmax id, max prob) pairs are already known (fused CTC head):
    same state lists, same result."""
    if last_max_prob_list is None:
        last_max_prob_list = []
    if last_max_index_list is None:
        last_max_index_list = []
    last_max_prob_list.extend(int(i) for i in ids)
    last_max_index_list.extend(np.float32(p) for i, p in zip(ids, max_probs) if i != blank_index)
    return decode_history(last_max_prob_list, last_max_index_list, vocabulary, blank_index) + \
        (last_max_prob_list, last_max_index_list)


def decode_history(index_history, prob_history, vocabulary, blank_index=0):
    """Collapse a full index history + score a non-blank prob history on the GPU."""
    eng = runtime.aux_engine()
    T = len(index_history)
    if T == 0:
        return 0, ''
    idx = torch.tensor(index_history, dtype=torch.int32, device=eng.device)
    # the collapse kernel scores every non-blank frame; feed the stored probs back at those frames
    mp = torch.zeros(T, dtype=torch.float32, device=eng.device)
    nb = idx != blank_index
    if len(prob_history):
        mp[nb] = torch.tensor(np.asarray(prob_history, np.float32), device=eng.device)
    return _collapse(eng, idx, mp, vocabulary, blank_index)
