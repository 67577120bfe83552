"""Several libraries across several GPUs: one process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" for CPU rehearsal), libraries dealt round-robin,
one integer all-reduce of the samples x tags matrix at the end.

The result equals `combineReadCounts` of the reference (tagdigger_fun.py:1061-1098) applied to
the per-file matrices -- files in sorted order, samples in order of first appearance, equal
sample names summed -- whatever the number of ranks, because integer addition is associative.
"""
import torch
import torch.distributed as dist


def sample_rows(bckeys):
    """Global sample order exactly as combineReadCounts builds it, and per file the row of each barcode."""
    order, slot, rows = [], {}, {}
    for f in sorted(bckeys.keys()):
        rows[f] = []
        for sample in bckeys[f][1]:
            if sample not in slot:
                slot[sample] = len(order)
                order.append(sample)
            rows[f].append(slot[sample])
    return order, rows


def count_libraries(bckeys, tags, cutsite, counter=None, device=None):
    """Every rank calls this with the same arguments.  `counter(file, barcodes, tags, cutsite)`
    returns the per-barcode matrix of one file (default: the GPU find_tags_fastq on this rank's
    device).  Returns [sample names, samples x tags counts] on every rank."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    if counter is None:
        from . import tagdigger_fun
        dev_index = device.index if isinstance(device, torch.device) and device.index is not None else 0

        def counter(f, barcodes, tgs, cs):
            return tagdigger_fun.find_tags_fastq(f, barcodes, tgs, cutsite=cs, device=dev_index)
    order, rows = sample_rows(bckeys)
    total = torch.zeros((len(order), len(tags)), dtype=torch.int64, device=device if device is not None else "cpu")
    for k, f in enumerate(sorted(bckeys.keys())):
        if k % world != rank:
            continue
        m = torch.tensor(counter(f, bckeys[f][0], tags, cutsite), dtype=torch.int64).reshape(len(bckeys[f][0]), len(tags))
        total.index_add_(0, torch.tensor(rows[f], dtype=torch.int64, device=total.device), m.to(total.device))
    if world > 1:
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
    return [order, total.cpu().tolist()]
