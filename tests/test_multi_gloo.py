"""The N > 1 path rehearsed on CPU: two processes over gloo, libraries dealt across ranks, one
all-reduce -- must equal combineReadCounts over the per-file matrices.  The CPU oracle stands in
for the per-rank GPU counter (there is no GPU here); the sharding and the reduction are the code
under test."""
import os
import random
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_inputs(tmpdir):
    from helpers import dirty_fastq, small_index
    rnd = random.Random(42)
    _, tags, cutsites = small_index(rnd, "TGCAG", nbar=1, ntag=24)
    bckeys = {}
    for k, name in enumerate(["libC.fq", "libA.fq", "libB.fq"]):
        barcodes, _, _ = small_index(rnd, "TGCAG", nbar=4 + k, ntag=1)
        data = dirty_fastq(rnd, barcodes, tags, cutsites, nrec=300)
        path = os.path.join(tmpdir, name)
        open(path, "wb").write(data)
        samples = ["s%d" % ((i * 2 + k) % 5) for i in range(len(barcodes))]   # names shared across files
        bckeys[path] = [barcodes, samples]
    return bckeys, tags


def _oracle_counter(f, barcodes, tags, cutsite):
    from oracle import c_oracle
    return c_oracle.find_tags_fastq(f, barcodes, tags, cutsite=cutsite)


def _worker(rank, world, port, bckeys, tags, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tagdigger_amd import multi
    res = multi.count_libraries(bckeys, tags, "TGCAG", counter=_oracle_counter)
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_combine_read_counts(tmp_path):
    from tagdigger_amd import tagdigger_fun as tf
    bckeys, tags = _make_inputs(str(tmp_path))
    countsdict = {f: _oracle_counter(f, bckeys[f][0], tags, "TGCAG") for f in bckeys}
    want = tf.combineReadCounts(countsdict, bckeys)
    assert sum(map(sum, want[1])) > 100
    out = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(2, _free_port(), bckeys, tags, out), nprocs=2, join=True)
    got = torch.load(out)
    assert got == want


def test_single_process_path(tmp_path):
    from tagdigger_amd import multi, tagdigger_fun as tf
    bckeys, tags = _make_inputs(str(tmp_path))
    countsdict = {f: _oracle_counter(f, bckeys[f][0], tags, "TGCAG") for f in bckeys}
    assert multi.count_libraries(bckeys, tags, "TGCAG", counter=_oracle_counter) == tf.combineReadCounts(countsdict, bckeys)
