"""otter_amd/generation.py (greedy / beam search / logits processors behind OtterForConditionalGeneration.generate) pinned
against the third-party implementation the reference delegates to: `transformers`' own generate() on a tiny LLaMA whose weights
are shared with otter_amd's LLaMA host (CPU, fp32).  The reference pins transformers==4.35.1; the installed version runs the
same algorithm (beam scorer with length penalty, n-gram / bad-word / min-length processors), which is what the call sites of
pipeline/demos and pipeline/benchmarks rely on."""
import pytest
import torch


@pytest.fixture(scope="module")
def pair():
    from transformers import LlamaConfig
    from transformers import LlamaForCausalLM as HFLlama

    from otter_amd.llama import LlamaForCausalLM

    cfg = LlamaConfig(hidden_size=48, intermediate_size=96, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
                      vocab_size=31, max_position_embeddings=128, rms_norm_eps=1e-6, tie_word_embeddings=False)
    torch.manual_seed(5)
    ref = HFLlama(cfg).eval()
    with torch.no_grad():
        for p in ref.parameters():
            p.mul_(3.0)          # sharper next-token distributions: beams diverge, eos shows up
    mine = LlamaForCausalLM(cfg).eval()
    mine.load_state_dict(ref.state_dict(), strict=False)
    return cfg, ref, mine


def _step_for(model, use_cache):
    def step(ids, mask, past, beam_idx):
        if use_cache and past is not None:
            if beam_idx is not None:
                past = tuple(tuple(t.index_select(0, beam_idx) for t in layer) for layer in past)
            out = model(input_ids=ids[:, -1:], attention_mask=mask, past_key_values=past, use_cache=True)
        else:
            out = model(input_ids=ids, attention_mask=mask, use_cache=use_cache)
        return out.logits[:, -1, :], (out.past_key_values if use_cache else None)
    return step


CASES = [
    dict(num_beams=1, do_sample=False, max_new_tokens=12),
    dict(num_beams=1, do_sample=False, max_new_tokens=12, no_repeat_ngram_size=2, repetition_penalty=1.3),
    dict(num_beams=3, do_sample=False, max_new_tokens=14, no_repeat_ngram_size=3),
    dict(num_beams=3, do_sample=False, max_new_tokens=14, no_repeat_ngram_size=3, bad_words_ids=[[7], [3, 4], [11, 2, 9]], length_penalty=0.6),
    dict(num_beams=4, do_sample=False, max_new_tokens=10, min_new_tokens=6, length_penalty=1.5, early_stopping=True),
    dict(num_beams=2, do_sample=False, max_new_tokens=9, num_return_sequences=2),
]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("use_cache", [False, True])
def test_matches_transformers_generate(pair, case, use_cache):
    from otter_amd.generation import generate_tokens

    cfg, ref, mine = pair
    kw = dict(CASES[case])
    g = torch.Generator().manual_seed(100 + case)
    ids = torch.randint(3, cfg.vocab_size, (3, 6), generator=g)
    mask = torch.ones_like(ids)
    eos, pad = 1, 0
    with torch.no_grad():
        want = ref.generate(input_ids=ids, attention_mask=mask, eos_token_id=eos, pad_token_id=pad, use_cache=True, **kw)
        got = generate_tokens(_step_for(mine, use_cache), ids, mask, eos_token_id=eos, pad_token_id=pad, **kw)
    assert got.shape == want.shape and torch.equal(got, want), (kw, got.tolist(), want.tolist())


def test_eos_is_actually_exercised(pair):
    """Sanity of the fixture: with these weights some greedy / beam outputs end in eos before max_new_tokens (so the
    finished-hypothesis branches above are not vacuous)."""
    from otter_amd.generation import generate_tokens

    cfg, _, mine = pair
    hit = False
    for seed in range(6):
        ids = torch.randint(3, cfg.vocab_size, (4, 5), generator=torch.Generator().manual_seed(seed))
        out = generate_tokens(_step_for(mine, True), ids, torch.ones_like(ids), eos_token_id=1, pad_token_id=0, num_beams=3, max_new_tokens=25)
        hit = hit or bool((out[:, 5:] == 1).any())
    assert hit


def test_sampling_is_reproducible_and_respects_top_k(pair):
    from otter_amd.generation import generate_tokens

    cfg, _, mine = pair
    ids = torch.randint(3, cfg.vocab_size, (2, 5), generator=torch.Generator().manual_seed(9))
    a = generate_tokens(_step_for(mine, True), ids, None, eos_token_id=1, max_new_tokens=8, do_sample=True, temperature=0.7, top_k=1,
                        generator=torch.Generator().manual_seed(1))
    greedy = generate_tokens(_step_for(mine, True), ids, None, eos_token_id=1, max_new_tokens=8)
    assert torch.equal(a, greedy)          # top_k = 1 sampling is greedy
    b = generate_tokens(_step_for(mine, True), ids, None, eos_token_id=1, max_new_tokens=8, do_sample=True, temperature=1.5, top_p=0.9,
                        generator=torch.Generator().manual_seed(2))
    c = generate_tokens(_step_for(mine, True), ids, None, eos_token_id=1, max_new_tokens=8, do_sample=True, temperature=1.5, top_p=0.9,
                        generator=torch.Generator().manual_seed(2))
    assert torch.equal(b, c)


def test_unsupported_arguments_are_loud(pair):
    from otter_amd.generation import generate_tokens

    _, _, mine = pair
    ids = torch.ones(1, 3, dtype=torch.long)
    with pytest.raises(NotImplementedError):
        generate_tokens(_step_for(mine, False), ids, None, max_new_tokens=2, num_beams=2, force_words_ids=[[5]])
    with pytest.raises(NotImplementedError):
        generate_tokens(_step_for(mine, False), ids, None, max_new_tokens=2, penalty_alpha=0.6, top_k=4)
    with pytest.raises(ValueError):      # GenerationConfig.validate's rules of the group mode
        generate_tokens(_step_for(mine, False), ids, None, max_new_tokens=2, num_beams=3, num_beam_groups=2, diversity_penalty=0.5)
    with pytest.raises(ValueError):
        generate_tokens(_step_for(mine, False), ids, None, max_new_tokens=2, num_beams=4, num_beam_groups=2)
    with pytest.raises(ValueError):
        generate_tokens(_step_for(mine, False), ids, None, max_new_tokens=2, num_beams=4, num_beam_groups=2, diversity_penalty=0.5, do_sample=True)


def test_beam_sample_properties(pair):
    """Beam-sample (num_beams > 1 with do_sample; transformers 4.35.1 generation/utils.py beam_sample: candidates DRAWN from
    softmax(warped log-probabilities + beam scores), then ranked).  The installed transformers draws the same candidates but ranks the finished
    ones by their position in the draw, so its final pick can differ from the pinned version's; what is pinned here: the run is a function of
    the generator, every emitted token lies in its beam's top-k support given its prefix, the n-gram rule holds, and with one beam the loop
    degenerates to plain sampling."""
    from otter_amd.generation import generate_tokens

    cfg, ref, mine = pair
    ids = torch.randint(3, cfg.vocab_size, (3, 6), generator=torch.Generator().manual_seed(17))
    mask = torch.ones_like(ids)
    kw = dict(eos_token_id=1, pad_token_id=0, num_beams=3, do_sample=True, temperature=0.9, top_k=4, max_new_tokens=9, no_repeat_ngram_size=2)
    a = generate_tokens(_step_for(mine, True), ids, mask, generator=torch.Generator().manual_seed(3), **kw)
    b = generate_tokens(_step_for(mine, False), ids, mask, generator=torch.Generator().manual_seed(3), **kw)
    c = generate_tokens(_step_for(mine, True), ids, mask, generator=torch.Generator().manual_seed(4), **kw)
    assert torch.equal(a, b) and a.shape[0] == 3 and a.shape[1] <= 6 + 9
    assert not torch.equal(a, c)                       # another seed, another draw (3 sentences x 9 steps: a collision is not credible)
    with torch.no_grad():
        logits = mine(input_ids=a, attention_mask=torch.ones_like(a)).logits
    for r in range(a.shape[0]):
        seq = a[r].tolist()
        grams = set()
        for t in range(6, len(seq)):
            if seq[t] == 0 and 1 in seq[6:t]:          # padding after eos
                break
            lp = torch.log_softmax(logits[r, t - 1].float(), -1)
            bigrams = {(seq[i], seq[i + 1]) for i in range(t - 1)}
            banned = [v for v in range(cfg.vocab_size) if (seq[t - 1], v) in bigrams]
            lp[banned] = float("-inf")
            assert seq[t] in lp.topk(4).indices.tolist(), (r, t, seq)
            assert (seq[t - 1], seq[t]) not in grams
            grams.add((seq[t - 1], seq[t]))
    # several sequences per sentence come back best first, as from beam search
    d = generate_tokens(_step_for(mine, True), ids, mask, generator=torch.Generator().manual_seed(3), num_return_sequences=2,
                        **{k: v for k, v in kw.items()})
    assert d.shape[0] == 6 and torch.equal(d[0::2][:, :a.shape[1]], a[:, :d.shape[1]])


@pytest.mark.parametrize("kw", [dict(num_beams=1, max_new_tokens=10), dict(num_beams=3, max_new_tokens=10, no_repeat_ngram_size=3)])
@pytest.mark.parametrize("use_cache", [False, True])
def test_left_padded_mixed_length_batch(pair, kw, use_cache):
    """the reference's benchmark wrappers tokenize with padding_side='left' (pipeline/benchmarks/models/otter_image.py), so a
    batch holds prompts of different lengths padded on the LEFT.  Greedy and beam search, with and without the KV cache, against
    transformers' generate on the same left-padded batch (position ids follow the mask there; RoPE is relative, so the tokens agree)."""
    from otter_amd.generation import generate_tokens

    cfg, ref, mine = pair
    g = torch.Generator().manual_seed(321)
    L = 9
    ids = torch.randint(3, cfg.vocab_size, (3, L), generator=g)
    mask = torch.ones_like(ids)
    for row, n_pad in enumerate([0, 3, 5]):
        ids[row, :n_pad] = 0
        mask[row, :n_pad] = 0
    with torch.no_grad():
        want = ref.generate(input_ids=ids, attention_mask=mask, eos_token_id=1, pad_token_id=0, use_cache=True, do_sample=False, **kw)
        got = generate_tokens(_step_for(mine, use_cache), ids, mask, eos_token_id=1, pad_token_id=0, **kw)
    assert torch.equal(got, want), (got.tolist(), want.tolist())


def test_llama_use_cache_follows_config(pair):
    """`past_key_values` come back whenever config.use_cache is true, in train mode too (HF / the reference's default)."""
    cfg, _, mine = pair
    ids = torch.randint(3, cfg.vocab_size, (2, 5), generator=torch.Generator().manual_seed(4))
    try:
        mine.train()
        with torch.no_grad():
            assert bool(cfg.use_cache) and mine(input_ids=ids).past_key_values is not None
            assert mine(input_ids=ids, use_cache=False).past_key_values is None
    finally:
        mine.eval()
    mine.release_fused_copies()      # idempotent on a model that never built any


@pytest.mark.parametrize("use_cache", [False, True])
def test_prefix_allowed_tokens_fn_matches_transformers(pair, use_cache):
    """PrefixConstrainedLogitsProcessor (the installed transformers still carries it): greedy and beam search under a per-sentence rule."""
    from otter_amd.generation import generate_tokens

    cfg, ref, mine = pair
    ids = torch.randint(3, cfg.vocab_size, (3, 5), generator=torch.Generator().manual_seed(31))
    mask = torch.ones_like(ids)

    def allowed(sentence, sofar):
        last = int(sofar[-1])
        return [v for v in range(2, cfg.vocab_size) if (v + last + sentence) % 3 != 0] + [1]

    for kw in (dict(num_beams=1, max_new_tokens=9), dict(num_beams=3, max_new_tokens=9, no_repeat_ngram_size=3)):
        with torch.no_grad():
            want = ref.generate(input_ids=ids, attention_mask=mask, eos_token_id=1, pad_token_id=0, do_sample=False, prefix_allowed_tokens_fn=allowed, **kw)
            got = generate_tokens(_step_for(mine, use_cache), ids, mask, eos_token_id=1, pad_token_id=0, prefix_allowed_tokens_fn=allowed, **kw)
        # (the installed transformers leaves eos instead of pad behind a finished hypothesis when this processor is on -- its running beams go
        #  on under the rule; 4.35.1's scorer pads: compare up to the first eos, and require padding behind it)
        assert got.shape == want.shape
        for r in range(got.shape[0]):
            g_, w_ = got[r, 5:].tolist(), want[r, 5:].tolist()
            n = g_.index(1) + 1 if 1 in g_ else len(g_)
            assert g_[:n] == w_[:n] and all(t == 0 for t in g_[n:]), (kw, r, g_, w_)
        new = got[:, 5:].tolist()
        for r, row in enumerate(new):
            prev = int(ids[r, -1])
            for t in row:
                if t == 0:
                    break
                assert t == 1 or (t + prev + r) % 3 != 0
                prev = t


def test_group_beam_search_with_identical_groups_is_beam_search(pair):
    """Diverse beam search (transformers 4.35.1 group_beam_search; removed from the installed transformers, so pinned through its defining
    properties): with the penalty at zero every group is an independent beam search of nb / ng beams over the same scores -- the best
    hypothesis is the one `num_beams = nb / ng` finds, and the groups return the same hypotheses."""
    from otter_amd.generation import generate_tokens

    cfg, _, mine = pair
    ids = torch.randint(3, cfg.vocab_size, (3, 6), generator=torch.Generator().manual_seed(41))
    mask = torch.ones_like(ids)
    common = dict(eos_token_id=1, pad_token_id=0, max_new_tokens=11, no_repeat_ngram_size=3, length_penalty=0.8)
    for use_cache in (False, True):
        plain = generate_tokens(_step_for(mine, use_cache), ids, mask, num_beams=2, num_return_sequences=2, **common)
        grp = generate_tokens(_step_for(mine, use_cache), ids, mask, num_beams=6, num_beam_groups=3, diversity_penalty=0.0,
                              _identical_groups_ok=True, num_return_sequences=6, **common)
        L = max(plain.shape[1], grp.shape[1])
        pad = lambda t: torch.nn.functional.pad(t, (0, L - t.shape[1]))
        plain, grp = pad(plain), pad(grp)
        for b in range(3):
            mine_rows = {tuple(r.tolist()) for r in grp[6 * b: 6 * b + 6]}
            assert mine_rows == {tuple(r.tolist()) for r in plain[2 * b: 2 * b + 2]}, (b, grp[6 * b: 6 * b + 6].tolist(), plain[2 * b: 2 * b + 2].tolist())
            assert torch.equal(grp[6 * b], plain[2 * b])          # best first


@pytest.mark.parametrize("use_cache", [False, True])
def test_group_beam_search_hamming_diversity_against_a_direct_restatement(pair, use_cache):
    """With one beam per group the algorithm is a chain of greedy decoders, group g choosing argmax(log p - penalty x [how many of the groups
    before it took that token at this step]); the scores it accumulates include the penalty (the processed scores are what the beam scorer
    adds up).  Restated here directly on the host model and compared token for token, hypothesis order included."""
    from otter_amd.generation import generate_tokens

    cfg, _, mine = pair
    ng, new, pen = 4, 8, 1.7
    ids = torch.randint(3, cfg.vocab_size, (2, 5), generator=torch.Generator().manual_seed(43))
    got = generate_tokens(_step_for(mine, use_cache), ids, torch.ones_like(ids), eos_token_id=None, pad_token_id=0, max_new_tokens=new,
                          num_beams=ng, num_beam_groups=ng, diversity_penalty=pen, num_return_sequences=ng, length_penalty=1.0)
    assert got.shape == (2 * ng, 5 + new)
    for b in range(2):
        seqs = [ids[b].clone() for _ in range(ng)]
        tot = [0.0] * ng
        for _ in range(new):
            taken = []
            for g in range(ng):
                with torch.no_grad():
                    lp = torch.log_softmax(mine(input_ids=seqs[g][None]).logits[0, -1].float(), -1)
                for t in taken:
                    lp[t] -= pen
                t = int(lp.argmax())
                tot[g] += float(lp[t])
                taken.append(t)
                seqs[g] = torch.cat([seqs[g], torch.tensor([t])])
        order = sorted(range(ng), key=lambda g: -tot[g] / new)
        want = torch.stack([seqs[g] for g in order])
        assert torch.equal(got[ng * b: ng * b + ng], want), (b, got[ng * b: ng * b + ng].tolist(), want.tolist(), tot)
        first = [int(s[5]) for s in seqs]
        assert len(set(first)) > 1        # the penalty did push groups apart
