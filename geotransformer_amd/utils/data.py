"""Mirror of geotransformer/utils/data.py:13-189 (stack-mode pyramid + registration collate), device resident.

`precompute_data_stack_mode` keeps the reference's signature and output schema
({'points','lengths','neighbors','subsampling','upsampling'}: lists of tensors) but runs every
grid-subsample and radius search on the GPU.  One uniform grid is built per stage and reused by the
three searches that target that stage (self / sub-sampling / up-sampling all use radius r_i against
stage i), instead of the reference's ten independent kd-tree builds.
"""
import numpy as np
import torch

from .. import _lib, ext


def precompute_data_stack_mode(points, lengths, num_stages, voxel_size, radius, neighbor_limits, exact_width=True,
                               tie_order='canonical'):
    """Pyramid of points + neighbour index tensors (data.py:13-77).

    Args mirror the reference.  ``points``/``lengths`` may be CPU or device tensors; outputs live on the
    same device.  ``exact_width=True`` reproduces the reference's column count
    ``min(max_count, neighbor_limit)`` (needs one host read per search); ``exact_width=False`` always
    emits ``neighbor_limit`` columns (extra columns hold the pad index, which every consumer treats as
    "no neighbour") and never synchronises on the search results.  ``tie_order='reference'`` (with exact_width=True)
    orders equal-distance neighbours exactly like the reference's nanoflann + std::sort (validation mode for quantised
    real scans; the default 'canonical' orders ties by index).
    """
    assert num_stages == len(neighbor_limits)
    assert tie_order in ('canonical', 'reference') and (tie_order == 'canonical' or exact_width)
    _lib.require_gpu()
    home = points.device
    dev = home if home.type == 'cuda' else torch.device('cuda', torch.cuda.current_device())
    points = points.to(dev).contiguous()
    lengths = lengths.to(dev).contiguous()

    points_list, lengths_list, lengths_host = [], [], []
    for i in range(num_stages):
        if i > 0:
            buf, lengths = ext.grid_subsample_device(points, lengths, voxel_size)
            host = lengths.tolist()  # row counts are data dependent: the one host read per stage
            points = buf[: sum(host)]
        else:
            host = lengths.tolist()
        points_list.append(points)
        lengths_list.append(lengths)
        lengths_host.append(host)
        voxel_size *= 2

    grids = []
    r = radius
    for i in range(num_stages):
        if tie_order == 'reference':  # one nanoflann-shaped tree per stage cloud, shared by the searches against it
            grids.append((ext.KdTreeIndex(points_list[i], lengths_list[i]), r))
        else:
            grids.append(ext.RadiusGrid(points_list[i], lengths_list[i], r))
        r *= 2

    def search(grid, q_points, q_lengths, limit):
        if tie_order == 'reference':
            index, r_stage = grid
            return index.query(q_points, q_lengths, r_stage, limit=limit if limit > 0 else None)
        if exact_width or limit <= 0:
            _, max_count = grid.count(q_points, q_lengths)
            max_count = int(max_count.item())
            width = max_count if limit <= 0 else min(max_count, limit)
            return grid.query(q_points, q_lengths, width, row_capacity=max(max_count, 64))
        return grid.query(q_points, q_lengths, limit, overflow=overflow)

    overflow = torch.zeros(1, dtype=torch.int32, device=dev)
    neighbors_list, subsampling_list, upsampling_list = [], [], []
    for i in range(num_stages):
        neighbors_list.append(search(grids[i], points_list[i], lengths_list[i], neighbor_limits[i]))
        if i < num_stages - 1:
            subsampling_list.append(search(grids[i], points_list[i + 1], lengths_list[i + 1], neighbor_limits[i]))
            upsampling_list.append(search(grids[i + 1], points_list[i], lengths_list[i], neighbor_limits[i + 1]))
    if not exact_width:
        worst = int(overflow.item())
        if worst > 0:
            raise RuntimeError(f'radius search row capacity exceeded ({worst} neighbours in one ball); '
                               f'use exact_width=True for such dense clouds')

    def back(ts):
        return [t.to(home) for t in ts]

    return {
        'points': back(points_list),
        'lengths': back(lengths_list),
        'neighbors': back(neighbors_list),
        'subsampling': back(subsampling_list),
        'upsampling': back(upsampling_list),
        'lengths_host': lengths_host,  # extra key (python ints): lets the model slice clouds without device reads
    }


def registration_collate_fn_stack_mode(data_dicts, num_stages, voxel_size, search_radius, neighbor_limits,
                                       precompute_data=True, device=None, exact_width=True, tie_order='canonical'):
    """Registration collate in stack mode (data.py:139-189): [ref_1..ref_B, src_1..src_B] stacking.

    ``device`` (optional): place the stacked cloud on that device before the pyramid is built, so the
    returned dict is already device resident (the reference does this afterwards with ``to_cuda``).
    """
    batch_size = len(data_dicts)
    collated = {}
    for data_dict in data_dicts:
        for key, value in data_dict.items():
            if isinstance(value, np.ndarray):
                value = torch.from_numpy(value)
            collated.setdefault(key, []).append(value)

    feats = torch.cat(collated.pop('ref_feats') + collated.pop('src_feats'), dim=0)
    points_list = collated.pop('ref_points') + collated.pop('src_points')
    lengths = torch.LongTensor([p.shape[0] for p in points_list])
    points = torch.cat(points_list, dim=0)
    if device is not None:
        feats, points, lengths = feats.to(device), points.to(device), lengths.to(device)

    if batch_size == 1:
        for key, value in collated.items():
            collated[key] = value[0].to(device) if device is not None and torch.is_tensor(value[0]) else value[0]

    collated['features'] = feats
    if precompute_data:
        collated.update(precompute_data_stack_mode(points, lengths, num_stages, voxel_size, search_radius,
                                                   neighbor_limits, exact_width=exact_width, tie_order=tie_order))
    else:
        collated['points'] = points
        collated['lengths'] = lengths
    collated['batch_size'] = batch_size
    return collated


def calibrate_neighbors_stack_mode(dataset, collate_fn, num_stages, voxel_size, search_radius, keep_ratio=0.8,
                                   sample_threshold=2000):
    """Neighbour-limit calibration (mirror of geotransformer/utils/data.py:192-217) on the GPU.

    Same result as the reference: per stage, the histogram of self-search neighbour counts (clipped to the theoretical
    bound hist_n = ceil(4/3 pi (r/v + 1)^3)) is accumulated over dataset items until every stage has more than
    `sample_threshold` samples, and the limit is the number of histogram bins below `keep_ratio` of the mass.
    The reference materialises (N, hist_n) neighbour tables to count them; here `geotr_radius_count` counts directly.
    `collate_fn` is accepted for signature compatibility; items may be registration dicts (ref_points / src_points) or
    single-cloud dicts (points).
    """
    del collate_fn
    _lib.require_gpu()
    dev = torch.device('cuda', torch.cuda.current_device())
    hist_n = int(np.ceil(4 / 3 * np.pi * (search_radius / voxel_size + 1) ** 3))
    neighbor_hists = np.zeros((num_stages, hist_n), dtype=np.int64)
    for i in range(len(dataset)):
        item = dataset[i]
        clouds = [item['ref_points'], item['src_points']] if 'ref_points' in item else [item['points']]
        clouds = [torch.from_numpy(c) if isinstance(c, np.ndarray) else c for c in clouds]
        points = torch.cat([c.float() for c in clouds], dim=0).to(dev).contiguous()
        lengths = torch.tensor([c.shape[0] for c in clouds], dtype=torch.int64, device=dev)
        v, r = voxel_size, search_radius
        for s in range(num_stages):
            if s > 0:
                buf, lengths = ext.grid_subsample_device(points, lengths, v)
                points = buf[: int(lengths.sum().item())]
            v *= 2
            counts, _ = ext.RadiusGrid(points, lengths, r).count(points, lengths)
            r *= 2
            c = torch.clamp(counts.long(), max=hist_n)  # a (N, hist_n) table holds at most hist_n valid entries per row
            neighbor_hists[s] += torch.bincount(c, minlength=hist_n + 1)[:hist_n].cpu().numpy()
        if np.min(np.sum(neighbor_hists, axis=1)) > sample_threshold:
            break
    cum_sum = np.cumsum(neighbor_hists.T, axis=0)
    neighbor_limits = np.sum(cum_sum < (keep_ratio * cum_sum[hist_n - 1, :]), axis=0)
    return neighbor_limits


def reset_seed_worker_init_fn(worker_id):
    """Seed numpy / random of a DataLoader worker from torch's per-worker seed (geotransformer/utils/torch.py:40-45)."""
    import random
    seed = torch.initial_seed() % (2 ** 32)
    np.random.seed(seed)
    random.seed(seed)


def _identity(batch):
    return batch


class StackModeLoader:
    """Iterable over collated stack-mode batches.  The torch DataLoader underneath (and its worker processes) only produce lists of
    host item dicts -- file IO and numpy augmentation; the collate, which builds the neighbour pyramid ON THE DEVICE, runs here in the
    consuming process as each batch is drawn, so forked workers never touch the GPU."""

    def __init__(self, loader, collate):
        self.loader = loader
        self.collate = collate
        self.dataset = loader.dataset
        self.sampler = loader.sampler  # DistributedSampler.set_epoch stays reachable, as trainers expect
        self.batch_size = loader.batch_size

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for items in self.loader:
            yield self.collate(items)


def build_dataloader_stack_mode(dataset, collate_fn, num_stages, voxel_size, search_radius, neighbor_limits, batch_size=1,
                                num_workers=1, shuffle=False, drop_last=False, distributed=False, precompute_data=True,
                                device=None, **collate_kwargs):
    """Mirror of geotransformer/utils/data.py:220-250 (+ utils/torch.py:48-77): same arguments, same batching / sampling / worker
    seeding; returns an iterable of collated batches.  `device` (default: the current HIP device when there is one) is where the
    collate places the stacked clouds and builds the pyramid."""
    from functools import partial
    if device is None and torch.cuda.is_available():
        device = torch.device('cuda', torch.cuda.current_device())
    sampler = torch.utils.data.DistributedSampler(dataset) if distributed else None
    loader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, num_workers=num_workers,
                                         shuffle=False if distributed else shuffle, sampler=sampler, collate_fn=_identity,
                                         worker_init_fn=reset_seed_worker_init_fn, pin_memory=False, drop_last=drop_last)
    if collate_fn is registration_collate_fn_stack_mode:
        collate_kwargs = dict(collate_kwargs, device=device)
    collate = partial(collate_fn, num_stages=num_stages, voxel_size=voxel_size, search_radius=search_radius,
                      neighbor_limits=neighbor_limits, precompute_data=precompute_data, **collate_kwargs)
    return StackModeLoader(loader, collate)
