"""Token generation for `OtterForConditionalGeneration.generate` (modeling_otter.py:999-1042).

The reference forwards its kwargs to the third-party `GenerationMixin.generate` of the pinned transformers==4.35.1 (not
importable with the transformers installed here).  What its own call sites use is restated here:
greedy search, beam search (`num_beams=3, no_repeat_ngram_size=3, bad_words_ids=...`: pipeline/demos/interactive/*.py,
pipeline/benchmarks/models/otter_{image,video}.py; `length_penalty`, `min_new_tokens`: the benchmark wrappers),
temperature / top-k / top-p sampling (pipeline/demos/demo_models.py:64-71,120-130; the serve UI's `do_sample` checkbox,
pipeline/serve/gradio_web_server.py:362-371) and -- no call site of the reference combines them, restated so that the two switches compose
as they do in GenerationMixin -- beam-sample (`num_beams > 1` with `do_sample`), diverse (group) beam search (`num_beam_groups > 1` with
`diversity_penalty`: generation/utils.py group_beam_search + HammingDiversityLogitsProcessor, round 6) and `prefix_allowed_tokens_fn`.  Not
restated: constrained beam search (`constraints` / `force_words_ids`), contrastive search, assisted decoding -- they raise.  The algorithm is transformers'
(generation/utils.py beam_search + generation/beam_search.py BeamSearchScorer + generation/logits_process.py), restated on
plain tensors around a `step` callback so that it serves both decoder hosts and both decode modes (KV cache or full
re-forward).  Pinned by tests/test_generation.py against `transformers`' own generate() on a shared tiny LLaMA.
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Sequence, Tuple

import torch

NEG_INF = float("-inf")


# ---- logits processors (generation/logits_process.py), applied in transformers' order ----------------------------------


def _repetition_penalty(ids: torch.Tensor, scores: torch.Tensor, penalty: float) -> torch.Tensor:
    s = torch.gather(scores, 1, ids)
    s = torch.where(s < 0, s * penalty, s / penalty)
    return scores.scatter(1, ids, s)


def _no_repeat_ngram(ids: torch.Tensor, scores: torch.Tensor, n: int) -> torch.Tensor:
    """Ban every token that would complete an n-gram already present in the row (prompt included)."""
    L = ids.shape[1]
    if n <= 0 or L + 1 < n:
        return scores
    rows = ids.tolist()
    for r, seq in enumerate(rows):
        if n == 1:
            banned = set(seq)
        else:
            prefix = tuple(seq[L - n + 1:])
            banned = {seq[i + n - 1] for i in range(L - n + 1) if tuple(seq[i:i + n - 1]) == prefix}
        if banned:
            scores[r, list(banned)] = NEG_INF
    return scores


def _no_bad_words(ids: torch.Tensor, scores: torch.Tensor, bad_words: Sequence[Sequence[int]]) -> torch.Tensor:
    single = [w[0] for w in bad_words if len(w) == 1]
    if single:
        scores[:, single] = NEG_INF
    multi = [w for w in bad_words if len(w) > 1]
    if multi:
        rows = ids.tolist()
        for r, seq in enumerate(rows):
            for w in multi:
                k = len(w) - 1
                if k <= len(seq) and seq[len(seq) - k:] == list(w[:-1]):
                    scores[r, w[-1]] = NEG_INF
    return scores


def _top_k_top_p(scores: torch.Tensor, top_k: int, top_p: float, min_keep: int = 1) -> torch.Tensor:
    if top_k and top_k > 0:
        k = min(max(top_k, min_keep), scores.shape[-1])
        kth = torch.topk(scores, k)[0][..., -1, None]
        scores = scores.masked_fill(scores < kth, NEG_INF)
    if top_p is not None and top_p < 1.0:
        sorted_s, sorted_i = torch.sort(scores, descending=False)
        cum = sorted_s.softmax(-1).cumsum(-1)
        remove = cum <= (1 - top_p)
        remove[..., -min_keep:] = False
        scores = scores.masked_fill(remove.scatter(1, sorted_i, remove), NEG_INF)
    return scores


class _Processors:
    def __init__(self, prompt_len: int, eos: List[int], repetition_penalty, no_repeat_ngram_size, bad_words_ids, min_new_tokens, min_length,
                 prefix_allowed_tokens_fn=None, beams_per_sentence: int = 1):
        self.prompt_len, self.eos = prompt_len, eos
        self.prefix_fn, self.per_sent = prefix_allowed_tokens_fn, max(1, int(beams_per_sentence))
        self.rp = float(repetition_penalty) if repetition_penalty not in (None, 1.0) else None
        self.ngram = int(no_repeat_ngram_size or 0)
        self.bad = [list(w) for w in (bad_words_ids or []) if list(w) != eos[:1] and not (len(w) == 1 and w[0] in eos)]
        self.min_new = int(min_new_tokens or 0)
        self.min_len = int(min_length or 0)

    def __call__(self, ids: torch.Tensor, scores: torch.Tensor) -> torch.Tensor:
        if self.rp is not None:
            scores = _repetition_penalty(ids, scores, self.rp)
        if self.ngram > 0:
            scores = _no_repeat_ngram(ids, scores, self.ngram)
        if self.bad:
            scores = _no_bad_words(ids, scores, self.bad)
        if self.eos and (ids.shape[1] < self.min_len or ids.shape[1] - self.prompt_len < self.min_new):
            scores[:, self.eos] = NEG_INF
        if self.prefix_fn is not None:
            # PrefixConstrainedLogitsProcessor: fn(sentence index, tokens so far) -> the ids allowed next; rows are [sentence][beam]
            keep = torch.full_like(scores, NEG_INF)
            for r in range(ids.shape[0]):
                allowed = list(self.prefix_fn(r // self.per_sent, ids[r]))
                if not allowed:
                    raise ValueError("`prefix_allowed_tokens_fn` returned an empty list for sentence %d" % (r // self.per_sent))
                keep[r, allowed] = 0.0
            scores = scores + keep
        return scores


# ---- beam hypotheses (generation/beam_search.py) ---------------------------------------------------------------------


class _BeamHyps:
    def __init__(self, num_beams: int, length_penalty: float, early_stopping):
        self.num_beams, self.lp, self.early = num_beams, length_penalty, early_stopping
        self.beams: List[Tuple[float, torch.Tensor]] = []
        self.worst = 1e9

    def __len__(self):
        return len(self.beams)

    def add(self, hyp: torch.Tensor, sum_logprobs: float, generated_len: int):
        score = sum_logprobs / (generated_len ** self.lp)
        if len(self) < self.num_beams or score > self.worst:
            self.beams.append((score, hyp))
            if len(self) > self.num_beams:
                srt = sorted((s, i) for i, (s, _) in enumerate(self.beams))
                del self.beams[srt[0][1]]
                self.worst = srt[1][0]
            else:
                self.worst = min(score, self.worst)

    def is_done(self, best_sum_logprobs: float, cur_len: int, prompt_len: int, max_length: int) -> bool:
        if len(self) < self.num_beams:
            return False
        if self.early is True:
            return True
        if self.early is False:
            return self.worst >= best_sum_logprobs / (cur_len - prompt_len) ** self.lp
        # "never": the best score any continuation could still reach
        if self.lp > 0.0:
            return self.worst >= best_sum_logprobs / (max_length - prompt_len) ** self.lp
        return self.worst >= best_sum_logprobs / (cur_len - prompt_len) ** self.lp


# ---- the loop ----------------------------------------------------------------------------------------------------------

# step(ids [N,L], mask [N,L] or None, past, beam_idx [N] or None) -> (last-position logits [N,V], past)
StepFn = Callable[[torch.Tensor, Optional[torch.Tensor], object, Optional[torch.Tensor]], Tuple[torch.Tensor, object]]


@torch.no_grad()
def generate_tokens(step: StepFn, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, *, max_new_tokens: Optional[int] = None,
                    max_length: Optional[int] = None, eos_token_id=None, pad_token_id: Optional[int] = None, num_beams: int = 1,
                    num_return_sequences: int = 1, length_penalty: float = 1.0, early_stopping=False, no_repeat_ngram_size: int = 0,
                    bad_words_ids=None, min_new_tokens: int = 0, min_length: int = 0, repetition_penalty: float = 1.0,
                    do_sample: bool = False, temperature: float = 1.0, top_k: int = 50, top_p: float = 1.0,
                    generator: Optional[torch.Generator] = None, num_beam_groups: int = 1, diversity_penalty: float = 0.0,
                    prefix_allowed_tokens_fn=None, _identical_groups_ok: bool = False, **unused) -> torch.Tensor:
    """Returns input_ids with the generated tokens appended ([B * num_return_sequences, L + new], right-padded with
    pad_token_id after eos), exactly the tensor `GenerationMixin.generate` returns for a decoder-only model."""
    if unused:
        bad = sorted(k for k, v in unused.items() if v is not None and k not in ("use_cache", "return_dict_in_generate", "output_scores"))
        if bad:
            raise NotImplementedError("otter_amd.generate: unsupported generation arguments %s" % bad)
    if num_return_sequences != 1 and not (num_beams > 1 and num_return_sequences <= num_beams):
        raise NotImplementedError("num_return_sequences > 1 needs num_beams >= num_return_sequences (beam search)")
    num_beam_groups = int(num_beam_groups or 1)
    diversity_penalty = float(diversity_penalty or 0.0)
    if num_beam_groups > 1:    # GenerationConfig.validate's rules for the group mode
        if num_beams % num_beam_groups != 0:
            raise ValueError("`num_beams` (%d) should be divisible by `num_beam_groups` (%d)" % (num_beams, num_beam_groups))
        if do_sample:
            raise ValueError("`do_sample` must be False with `num_beam_groups` > 1 (diverse beam search does not sample)")
        if diversity_penalty <= 0.0 and not _identical_groups_ok:
            raise ValueError("`diversity_penalty` should be greater than 0.0 with `num_beam_groups` > 1, otherwise the groups will be identical")
    elif diversity_penalty != 0.0:
        raise ValueError("`diversity_penalty` is only used by group beam search: set `num_beam_groups` > 1")
    B, L0 = input_ids.shape
    dev = input_ids.device
    eos = [] if eos_token_id is None else ([int(eos_token_id)] if isinstance(eos_token_id, int) else [int(e) for e in eos_token_id])
    if pad_token_id is None:
        pad_token_id = eos[0] if eos else 0
    if max_new_tokens is not None:
        max_len = L0 + int(max_new_tokens)
    elif max_length is not None:
        max_len = int(max_length)
    else:
        max_len = L0 + 20
    proc = _Processors(L0, eos, repetition_penalty, no_repeat_ngram_size, bad_words_ids, min_new_tokens, min_length, prefix_allowed_tokens_fn,
                       beams_per_sentence=(num_beams // num_beam_groups if num_beam_groups > 1 else num_beams))
    if max_len <= L0:
        return input_ids
    eos_t = torch.tensor(eos, device=dev, dtype=torch.long) if eos else None

    if num_beams == 1:
        ids, mask, past = input_ids, attention_mask, None
        unfinished = torch.ones(B, dtype=torch.bool, device=dev)
        while ids.shape[1] < max_len:
            logits, past = step(ids, mask, past, None)
            scores = proc(ids, logits.float().clone())
            if do_sample:
                if temperature is not None and temperature != 1.0:
                    scores = scores / float(temperature)
                scores = _top_k_top_p(scores, int(top_k or 0), float(top_p if top_p is not None else 1.0))
                nxt = torch.multinomial(scores.softmax(-1), 1, generator=generator).squeeze(1)
            else:
                nxt = scores.argmax(-1)
            nxt = torch.where(unfinished, nxt, torch.full_like(nxt, pad_token_id))
            ids = torch.cat([ids, nxt[:, None]], dim=1)
            if mask is not None:
                mask = torch.cat([mask, torch.ones_like(mask[:, :1])], dim=1)
            if eos_t is not None:
                unfinished = unfinished & ~torch.isin(nxt, eos_t)
                if not bool(unfinished.any()):
                    break
        return ids

    if num_beam_groups > 1:
        return _group_beam_search(step, input_ids, attention_mask, proc, eos, pad_token_id, max_len, int(num_beams), num_beam_groups,
                                  diversity_penalty, float(length_penalty), early_stopping, int(num_return_sequences))

    # ---- beam search ----
    nb = int(num_beams)
    ids = input_ids.repeat_interleave(nb, dim=0)
    mask = attention_mask.repeat_interleave(nb, dim=0) if attention_mask is not None else None
    beam_scores = torch.zeros(B, nb, dtype=torch.float32, device=dev)
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view(-1)
    hyps = [_BeamHyps(nb, float(length_penalty), early_stopping) for _ in range(B)]
    done = [False] * B
    past, beam_idx = None, None
    while True:
        logits, past = step(ids, mask, past, beam_idx)
        logp = torch.log_softmax(logits.float(), dim=-1)
        logp = proc(ids, logp)
        V = logp.shape[-1]
        n_cand = max(2, 1 + len(eos)) * nb
        if do_sample:
            # beam-sample (generation/utils.py beam_sample): the warpers act on the processed log-probabilities of each beam (at least as many
            # tokens kept as candidates are drawn per beam pair), candidates are DRAWN from softmax(score + beam score) over all beams of a
            # sentence instead of taken by top-k, then ranked by that score
            if temperature is not None and temperature != 1.0:
                logp = logp / float(temperature)
            logp = _top_k_top_p(logp, int(top_k or 0), float(top_p if top_p is not None else 1.0), min_keep=max(2, 1 + len(eos)))
            scores = (logp + beam_scores[:, None]).view(B, nb * V)
            ni = torch.multinomial(scores.softmax(-1), n_cand, generator=generator)
            ns = scores.gather(1, ni)
            ns, order = torch.sort(ns, descending=True, dim=1)
            ni = ni.gather(1, order)
        else:
            scores = (logp + beam_scores[:, None]).view(B, nb * V)
            ns, ni = torch.topk(scores, n_cand, dim=1, largest=True, sorted=True)
        n_idx, n_tok = (ni // V).tolist(), (ni % V).tolist()
        ns_l = ns.tolist()
        cur_len = ids.shape[1] + 1
        new_scores = torch.zeros(B, nb, dtype=torch.float32)
        new_tokens = torch.full((B, nb), pad_token_id, dtype=torch.long)
        new_index = torch.zeros(B, nb, dtype=torch.long)
        for b in range(B):
            if done[b]:
                new_index[b] = b * nb          # padded beams of a finished sentence (transformers pads tokens, keeps index 0)
                continue
            k = 0
            for rank, (tok, sc, bi) in enumerate(zip(n_tok[b], ns_l[b], n_idx[b])):
                row = b * nb + bi
                if eos and tok in eos:
                    if rank >= nb:
                        continue
                    hyps[b].add(ids[row].clone(), sc, cur_len - L0)
                else:
                    new_scores[b, k], new_tokens[b, k], new_index[b, k] = sc, tok, row
                    k += 1
                if k == nb:
                    break
            done[b] = done[b] or hyps[b].is_done(max(ns_l[b]), cur_len, L0, max_len)
        beam_scores = new_scores.view(-1).to(dev)
        beam_idx = new_index.view(-1).to(dev)
        ids = torch.cat([ids[beam_idx], new_tokens.view(-1, 1).to(dev)], dim=1)
        if mask is not None:
            mask = torch.cat([mask[beam_idx], torch.ones_like(mask[:, :1])], dim=1)
        if all(done) or ids.shape[1] >= max_len:
            break
    # finalize: open beams of unfinished sentences become hypotheses, best `num_return_sequences` per sentence
    bs = beam_scores.tolist()
    for b in range(B):
        if done[b]:
            continue
        for j in range(nb):
            hyps[b].add(ids[b * nb + j], bs[b * nb + j], ids.shape[1] - L0)
    return _pick_best([hyps[b].beams for b in range(B)], num_return_sequences, max_len, pad_token_id, eos, dev)


def _pick_best(beams_per_sentence, num_return_sequences: int, max_len: int, pad_token_id: int, eos: List[int], dev) -> torch.Tensor:
    """BeamSearchScorer.finalize's selection: the best `num_return_sequences` hypotheses of every sentence, best first, right-padded."""
    best = []
    for beams in beams_per_sentence:
        srt = sorted(beams, key=lambda x: x[0])
        for _ in range(num_return_sequences):
            best.append(srt.pop()[1])
    sent_max = min(max(int(h.shape[0]) for h in best) + 1, max_len)
    out = torch.full((len(best), sent_max), pad_token_id, dtype=torch.long, device=dev)
    for i, h in enumerate(best):
        out[i, :h.shape[0]] = h
        if h.shape[0] < sent_max and eos:
            out[i, h.shape[0]] = eos[0]
    return out


def _group_beam_search(step: StepFn, input_ids, attention_mask, proc: _Processors, eos: List[int], pad_token_id: int, max_len: int, nb: int,
                       ng: int, diversity_penalty: float, length_penalty: float, early_stopping, num_return_sequences: int) -> torch.Tensor:
    """Diverse beam search (transformers 4.35.1 generation/utils.py group_beam_search; Vijayakumar et al. 2016): the `nb` beams of a sentence
    are `ng` groups of `nb / ng`; one decoder step serves all of them, then the groups choose ONE AFTER THE OTHER, each an ordinary beam-search
    step over its own beams whose log-probabilities were first lowered by `diversity_penalty` x (how many beams of the EARLIER groups of the
    same sentence chose that token at this step) -- HammingDiversityLogitsProcessor, applied before the other processors as in
    `_get_logits_processor`.  Hypotheses are kept per (sentence, group) with `nb / ng` slots each (BeamSearchScorer with num_beam_groups);
    the returned sequences are the best of the union of a sentence's groups."""
    B, L0 = input_ids.shape
    dev = input_ids.device
    gs = nb // ng
    ids = input_ids.repeat_interleave(nb, dim=0)
    mask = attention_mask.repeat_interleave(nb, dim=0) if attention_mask is not None else None
    beam_scores = torch.full((B, nb), -1e9, dtype=torch.float32)
    beam_scores[:, ::gs] = 0.0           # one live beam per group at the first step
    beam_scores = beam_scores.view(-1)
    hyps = [[_BeamHyps(gs, length_penalty, early_stopping) for _ in range(ng)] for _ in range(B)]
    done = [[False] * ng for _ in range(B)]
    past, reorder = None, None
    n_cand = max(2, 1 + len(eos)) * gs
    while True:
        logits, past = step(ids, mask, past, reorder)
        logp_all = torch.log_softmax(logits.float(), dim=-1)
        V = logp_all.shape[-1]
        cur_len = ids.shape[1] + 1
        current = torch.zeros(B * nb, dtype=torch.long)          # this step's choice of every beam (earlier groups filled in first)
        reorder_l = torch.arange(B * nb, dtype=torch.long)
        new_scores = beam_scores.clone()
        ids_cpu = ids.cpu()
        for g in range(ng):
            rows = torch.tensor([b * nb + g * gs + j for b in range(B) for j in range(gs)], dtype=torch.long)
            rows_d = rows.to(dev)
            gids = ids.index_select(0, rows_d)
            logp = logp_all.index_select(0, rows_d).clone()
            if g > 0 and diversity_penalty != 0.0:
                for b in range(B):
                    freq = torch.bincount(current[b * nb: b * nb + g * gs], minlength=V).to(logp.dtype).to(dev)
                    logp[b * gs:(b + 1) * gs] -= diversity_penalty * freq
            logp = proc(gids, logp)
            scores = (logp + beam_scores.index_select(0, rows).to(dev)[:, None]).view(B, gs * V)
            ns, ni = torch.topk(scores, n_cand, dim=1, largest=True, sorted=True)
            n_idx, n_tok, ns_l = (ni // V).tolist(), (ni % V).tolist(), ns.tolist()
            for b in range(B):
                base = b * nb + g * gs
                if done[b][g]:
                    new_scores[base: base + gs] = 0.0
                    current[base: base + gs] = pad_token_id
                    continue
                k = 0
                for rank, (tok, sc, bi) in enumerate(zip(n_tok[b], ns_l[b], n_idx[b])):
                    if eos and tok in eos:
                        if rank >= gs:
                            continue
                        hyps[b][g].add(ids_cpu[base + bi].clone().to(dev), sc, cur_len - L0)
                    else:
                        new_scores[base + k], current[base + k], reorder_l[base + k] = sc, tok, base + bi
                        k += 1
                    if k == gs:
                        break
                done[b][g] = done[b][g] or hyps[b][g].is_done(max(ns_l[b]), cur_len, L0, max_len)
        beam_scores = new_scores
        reorder = reorder_l.to(dev)
        ids = torch.cat([ids.index_select(0, reorder), current.to(dev)[:, None]], dim=1)
        if mask is not None:
            mask = torch.cat([mask.index_select(0, reorder), torch.ones_like(mask[:, :1])], dim=1)
        if all(all(d) for d in done) or ids.shape[1] >= max_len:
            break
    bs = beam_scores.tolist()
    for b in range(B):
        for g in range(ng):
            if done[b][g]:
                continue
            for j in range(gs):
                r = b * nb + g * gs + j
                hyps[b][g].add(ids[r], bs[r], ids.shape[1] - L0)
    return _pick_best([[h for g in range(ng) for h in hyps[b][g].beams] for b in range(B)], num_return_sequences, max_len, pad_token_id, eos, dev)
