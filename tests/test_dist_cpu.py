"""CPU tests of the N>1 path with the gloo backend, world_size 2 (the 8-GPU run is the driver's):
weight broadcast in flat buckets, contiguous prompt sharding, and the RNG contract under sharding
(per-rank noise slices concatenate to the single-process noise)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from audioldm2_amd import dist as adist
    from audioldm2_amd.ddim import DDIMSampler
    r, w, _ = adist.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    # 1) bucketed weight broadcast: rank 0's values arrive everywhere, packed caches are dropped
    torch.manual_seed(100 + rank)
    net = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.Conv2d(8, 8, 3), torch.nn.GroupNorm(4, 8))
    net[0]._pk = "stale"
    sent = adist.broadcast_tensors(list(net.parameters()), src=0, bucket_bytes=4096)  # forces several buckets
    adist.broadcast_module(net, src=0)
    assert sent == sum(p.numel() * 4 for p in net.parameters()) and net[0]._pk is None
    torch.manual_seed(100)
    ref = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.Conv2d(8, 8, 3), torch.nn.GroupNorm(4, 8))
    for a, b in zip(net.parameters(), ref.parameters()):
        assert torch.equal(a, b)
    # 2) sharding: contiguous, disjoint, covering
    lo, hi = adist.shard_range(7, rank, world)
    # 3) RNG contract: this rank's noise rows == rows [lo, hi) of the single-process draw

    class M:
        num_timesteps = 1000
        noise_shard = (7, lo)
    s = DDIMSampler(M())
    torch.manual_seed(42)
    img, noise, _ = s._draw_noise((hi - lo, 8, 4, 4), 3, None, False)
    torch.save({"lo": lo, "hi": hi, "img": img, "noise": noise}, os.path.join(out_dir, f"r{rank}.pt"))
    # 4) result gather in rank order
    import numpy as np
    got = adist.gather_waveforms(np.full((hi - lo, 1, 5), float(rank), dtype=np.float32), dst=0)
    if rank == 0:
        assert got.shape == (7, 1, 5) and got[:, 0, 0].tolist() == [0.0] * 4 + [1.0] * 3
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_broadcast_shard_and_noise_contract(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world)]
    assert (parts[0]["lo"], parts[0]["hi"], parts[1]["lo"], parts[1]["hi"]) == (0, 4, 4, 7)
    torch.manual_seed(42)
    full = [torch.randn(7, 8, 4, 4) for _ in range(4)]
    assert torch.equal(torch.cat([p["img"] for p in parts]), full[0])
    for i in range(3):
        assert torch.equal(torch.cat([p["noise"][i] for p in parts]), full[i + 1])


def test_shard_range_properties():
    from audioldm2_amd.dist import shard_range
    for n in (1, 7, 8, 64, 65):
        for w in (1, 2, 3, 8):
            rs = [shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1


def test_candidate_rows_keep_a_prompts_candidates_on_one_rank_and_rerank_like_one_process():
    """Prompt sharding with n_candidate_gen_per_text > 1: the reference lays the candidates out candidate-major
    (`torch.cat([z] * n_gen)`, row = candidate * B + prompt) and picks `best = i + argmax(similarity[i::B]) * B`
    (ddpm.py:1559-1564).  A rank samples every candidate of ITS prompts, draws the global noise and keeps those rows; ranking
    them locally must choose the very rows a single process would."""
    from audioldm2_amd.ddim import host_drawer
    from audioldm2_amd.dist import candidate_rows, shard_range
    B, n_gen = 7, 3
    g = torch.Generator().manual_seed(0)
    sim = torch.rand(B * n_gen, generator=g)                       # one similarity per global row
    single = [i + int(torch.argmax(sim[i::B])) * B for i in range(B)]
    for world in (1, 2, 3, 8):
        seen, chosen = [], []
        for rank in range(world):
            rows = candidate_rows(B, n_gen, rank, world)
            lo, hi = shard_range(B, rank, world)
            Bp = hi - lo
            assert rows.numel() == Bp * n_gen
            assert torch.equal(rows.view(n_gen, Bp) % B, torch.arange(lo, hi).expand(n_gen, Bp))   # our prompts, all candidates
            assert torch.equal(rows.view(n_gen, Bp) // B, torch.arange(n_gen)[:, None].expand(n_gen, Bp))  # candidate-major
            seen += rows.tolist()
            local = sim[rows]
            best = [j + int(torch.argmax(local[j::Bp])) * Bp for j in range(Bp)] if Bp else []
            chosen += [int(rows[b]) for b in best]
            # the noise rows of this rank = those rows of the single-process draw
            if Bp:
                torch.manual_seed(5)
                mine = host_drawer((Bp * n_gen, 2, 3), (B * n_gen, rows))()
                torch.manual_seed(5)
                full = torch.randn(B * n_gen, 2, 3)
                assert torch.equal(mine, full[rows])
                dst = torch.empty(Bp * n_gen, 2, 3)
                torch.manual_seed(5)
                host_drawer((Bp * n_gen, 2, 3), (B * n_gen, rows)).into(dst)
                assert torch.equal(dst, full[rows])
        assert sorted(seen) == list(range(B * n_gen))
        assert chosen == single
    # one candidate per prompt: the rows are the contiguous slice the n_gen == 1 path has always used
    assert candidate_rows(7, 1, 1, 2).tolist() == list(range(*shard_range(7, 1, 2)))


def test_sharded_reranking_consumes_the_global_unconditional_draws():
    """`cos_similarity` replaces each row's CLAP embedding by the empty-text embedding with probability 0.1 — one
    `torch.rand(1)` per row of the GLOBAL candidate batch (audio pass, then text pass; encoders/modules.py:728-735).  A prompt
    shard must draw the global batch's uniforms and apply its rows' decisions, or its candidates are ranked with other rows
    replaced than in the single-process run."""
    import types
    from audioldm2_amd.clap import CLAPAudioEmbeddingClassifierFreev2 as Clap
    from audioldm2_amd.dist import candidate_rows
    B, n_gen, world = 5, 3, 2
    G = B * n_gen

    def fake(decision_shard):
        ns = types.SimpleNamespace(unconditional_prob=0.5, unconditional_token=torch.full((1, 4), -7.0),
                                   decision_shard=decision_shard)
        ns.make_decision = lambda pr: float(torch.rand(1)) < pr
        return ns

    def two_passes(ns, emb):   # cos_similarity: audio embeddings, then text embeddings
        a = Clap._draw_unconditional(ns, emb.clone())
        t = Clap._draw_unconditional(ns, emb.clone() + 100.0)
        return a, t

    emb = torch.arange(G * 4, dtype=torch.float32).view(G, 4)
    torch.manual_seed(123)
    ga, gt = two_passes(fake(None), emb)
    assert (ga[:, 0, 0] == -7.0).any() and not (ga[:, 0, 0] == -7.0).all()   # the seed replaces some rows, not all
    end_state = torch.get_rng_state()
    for rank in range(world):
        rows = candidate_rows(B, n_gen, rank, world)
        torch.manual_seed(123)
        la, lt = two_passes(fake((G, rows.tolist())), emb[rows])
        assert torch.equal(la, ga[rows]) and torch.equal(lt, gt[rows])
        assert torch.equal(torch.get_rng_state(), end_state)   # and the generator ends where the single process leaves it


def _worker8(rank, world, port, out_dir, B, n_gen):
    """One rank of the 8-rank job below: everything a prompt shard does around the (GPU) sampler, on the CPU."""
    import types
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import numpy as np
    from audioldm2_amd import dist as adist
    from audioldm2_amd.clap import CLAPAudioEmbeddingClassifierFreev2 as Clap
    from audioldm2_amd.ddim import DDIMSampler
    torch.set_num_threads(1)
    r, w, _ = adist.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(200 + rank)
    net = torch.nn.Sequential(torch.nn.Linear(32, 32), torch.nn.GroupNorm(4, 8))
    adist.broadcast_module(net, src=0)
    rows = adist.candidate_rows(B, n_gen, rank, world)
    lo, hi = adist.shard_range(B, rank, world)
    Bp = hi - lo
    G = B * n_gen

    class M:
        num_timesteps = 1000
        noise_shard = (G, rows)
    torch.manual_seed(42)   # seed_everything(42) of every rank
    img, noise, _ = DDIMSampler(M())._draw_noise((Bp * n_gen, 4, 2, 2), 3, None, False)
    # re-ranking: the global batch's unconditional-probability draws (audio pass, then text pass), our rows' decisions
    ns = types.SimpleNamespace(unconditional_prob=0.5, unconditional_token=torch.full((1, 4), -7.0),
                               decision_shard=(G, rows.tolist()))
    ns.make_decision = lambda pr: float(torch.rand(1)) < pr
    emb = torch.arange(G * 4, dtype=torch.float32).view(G, 4)
    a = Clap._draw_unconditional(ns, emb[rows].clone())
    t = Clap._draw_unconditional(ns, emb[rows].clone() + 100.0)
    sim = torch.rand(G, generator=torch.Generator().manual_seed(9))[rows]          # the similarities this rank would compute
    best = [j + int(torch.argmax(sim[j::Bp])) * Bp for j in range(Bp)]
    wave = np.stack([np.full((1, 5), float(rows[b]), dtype=np.float32) for b in best]) if Bp else np.zeros((0, 1, 5), np.float32)
    got = adist.gather_waveforms(wave, dst=0)
    torch.save({"rows": rows, "img": img, "noise": noise, "a": a, "t": t, "w0": net[0].weight.detach().clone(),
                "gathered": got, "rng": torch.get_rng_state()}, os.path.join(out_dir, f"r{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_eight_rank_gloo_uneven_batch_with_candidates_equals_single_process(tmp_path):
    """the shape of the driver's 8-GPU run, on the CPU — 8 gloo ranks, an UNEVEN global batch (13 prompts: five
    ranks hold 2, three hold 1) and n_candidate_gen_per_text = 3.  Every rank's noise rows, re-ranker decisions and chosen
    candidates must be the single-process run's; the generator of every rank must end where the single process leaves it;
    the gather returns the prompts in global order; the broadcast delivers rank 0's weights."""
    import types
    from audioldm2_amd.clap import CLAPAudioEmbeddingClassifierFreev2 as Clap
    world, B, n_gen = 8, 13, 3
    G = B * n_gen
    port = _free_port()
    mp.spawn(_worker8, args=(world, port, str(tmp_path), B, n_gen), nprocs=world, join=True)
    parts = [torch.load(os.path.join(tmp_path, f"r{r}.pt"), weights_only=False) for r in range(world)]
    assert [p["rows"].numel() // n_gen for p in parts] == [2, 2, 2, 2, 2, 1, 1, 1]
    # single process: the same draws on the global batch
    torch.manual_seed(42)
    full = [torch.randn(G, 4, 2, 2) for _ in range(4)]
    ns = types.SimpleNamespace(unconditional_prob=0.5, unconditional_token=torch.full((1, 4), -7.0), decision_shard=None)
    ns.make_decision = lambda pr: float(torch.rand(1)) < pr
    emb = torch.arange(G * 4, dtype=torch.float32).view(G, 4)
    ga = Clap._draw_unconditional(ns, emb.clone())
    gt = Clap._draw_unconditional(ns, emb.clone() + 100.0)
    end_state = torch.get_rng_state()
    sim = torch.rand(G, generator=torch.Generator().manual_seed(9))
    single = [i + int(torch.argmax(sim[i::B])) * B for i in range(B)]
    for p in parts:
        rows = p["rows"]
        assert torch.equal(p["img"], full[0][rows])
        for i in range(3):
            assert torch.equal(p["noise"][i], full[i + 1][rows])
        assert torch.equal(p["a"], ga[rows]) and torch.equal(p["t"], gt[rows])
        assert torch.equal(p["rng"], end_state)
        assert torch.equal(p["w0"], parts[0]["w0"])
    got = parts[0]["gathered"]
    assert got.shape == (B, 1, 5) and got[:, 0, 0].astype(int).tolist() == single
    assert all(p["gathered"] is None for p in parts[1:])
