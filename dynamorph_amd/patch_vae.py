"""Latent-encoding driver: mirror of pipeline/patch_VAE.py::process_VAE (VQ branch, lines 343-462).

The reference encodes one patch at a time (batch-of-one `model.enc` -> `model.vq` calls with the model
left in train mode, two device->host copies per patch).  Here whole batches run through the HIP
encoder with PER-SAMPLE BatchNorm statistics, which is arithmetically the same thing
 and keeps samples independent -- the property that lets the path shard over GPUs
with no collective (dynamorph_amd.dist.shard_range).
"""
import os
import pickle

import numpy as np
import torch

from . import engine as E
from .train_utils import zscore_patch


def encode_patches(model, patches, device="cuda:0", batch_size=1024, zscore_on_device=False):
    """patches: (N, C, H, W) float tensor/array on the host.  Returns (z_before, z_after) as float32
    numpy arrays of shape (N, D*H/8*W/8), in input order (patch_VAE.py:454,459).
    zscore_on_device: `patches` are the RAW (float64) patches; each batch is z-scored per patch and channel on
    the GPU (dm_zscore_patch, double arithmetic) instead of on the host (patch_VAE.py:413-419)."""
    from . import ops
    patches = torch.as_tensor(patches)
    if patches.dim() != 4:
        raise AssertionError("dataset tensor dimension can only be 4, not {}".format(patches.dim()))
    from .vq_vae import VQ_VAE, VQ_VAE_z32
    if isinstance(model, VQ_VAE):
        layers = E.Layers(model)
        codebook = layers.codebook.weight
        e1 = None                           # composite first-layer weights: once per call (the weights do not change here)

        def encode(x):
            nonlocal e1
            if e1 is None:
                e1 = E.e1_operands(layers)
            z, cx = E.encoder_forward(layers, x, per_sample=True, e1=e1, join=False, latents_only=True)
            return z, cx.join
    elif isinstance(model, VQ_VAE_z32):
        enc = model.enc                     # children 0/1/3/4: conv, BatchNorm, conv, BatchNorm; 5: ResidualBlock
        codebook = model.vq.w.weight

        def encode(x):
            h, _ = E.z32_stem_forward(enc[0], enc[1], enc[3], enc[4], x, per_sample=True)
            return E.residual_forward(enc[5]._handles(), h, True)[0], None
    else:
        codebook = model.vq.w.weight

        def encode(x):                      # any other module: the reference's batch-of-one loop as it is
            return torch.cat([model.enc(x[j:j + 1]) for j in range(x.shape[0])], 0), None
    device = torch.device(device)
    N = patches.shape[0]
    if N == 0:
        return np.zeros((0, 0), np.float32), np.zeros((0, 0), np.float32)
    # Three stages in flight: batch i+1 crosses PCIe on a copy stream while batch i is encoded, and the latents of batch
    # i-1 go back on a second copy stream straight into the result arrays (a synchronous .cpu() per batch would leave the
    # GPU idle for both transfers: at 2 M patches/s one batch of 1024 is 0.5 ms of kernels against 134 MB in and 33 MB
    # out).  Measured on the MI355X host (tools/exp/host_alloc_probe.py): DMA from / to pinned memory 53-56 GB/s, from
    # pageable memory 10 GB/s, into freshly allocated pageable memory 5 GB/s (first-touch page faults).  Hence: hand
    # over PINNED patches for the full rate (process_VAE does: 365 k patches/s host to host, the PCIe limit); pageable
    # patches of the right dtype are copied by the runtime's own staging (this thread blocks, the queued kernels do
    # not: 120 k patches/s); patches that need a dtype conversion go through two pinned staging buffers.  The result
    # arrays are allocated pinned (25 GB/s to allocate) unless they exceed DM_PINNED_RESULT_BYTES (default 16 GiB; then
    # pageable, pre-faulted by a parallel fill, and written by blocking copies on a helper thread).
    from concurrent.futures import ThreadPoolExecutor
    bs = int(min(batch_size, N))
    in_dtype = patches.dtype if zscore_on_device else torch.float32
    pin_cap = int(os.environ.get("DM_PINNED_RESULT_BYTES", str(16 << 30)))
    res = [None, None]
    with torch.no_grad(), torch.cuda.device(device), ThreadPoolExecutor(1) as helper:   # (non-zero gpu ids: patch_VAE.py:422)
        compute = torch.cuda.current_stream()
        s_in, s_out = torch.cuda.Stream(), torch.cuda.Stream()
        shape = (bs,) + tuple(patches.shape[1:])
        x_dev = [torch.empty(shape, dtype=in_dtype, device=device) for _ in range(2)]
        convert = patches.dtype != in_dtype
        in_pin = [torch.empty(shape, dtype=in_dtype, pin_memory=True) for _ in range(2)] if convert else None
        ev_in = [torch.cuda.Event() for _ in range(2)]          # the batch has reached x_dev[k]
        ev_done = [torch.cuda.Event() for _ in range(2)]        # the kernels reading x_dev[k] have finished
        sent = [None, None]                                     # helper's future for the batch that last used slot k

        def hand_back(lo, n, z_b, z_a, ev):
            with torch.cuda.device(device), torch.cuda.stream(s_out):
                s_out.wait_event(ev)
                res[0][lo:lo + n].copy_(z_b, non_blocking=True)     # asynchronous into pinned results; into pageable ones
                res[1][lo:lo + n].copy_(z_a, non_blocking=True)     # it blocks, but only this helper thread

        for it, lo in enumerate(range(0, N, bs)):
            k = it & 1
            n = min(bs, N - lo)
            if sent[k] is not None:
                sent[k].result()                                # at most two batches of latents wait on the device
            src = patches[lo:lo + n]
            if convert:
                ev_in[k].synchronize()                          # staging buffer k has left for the device
                in_pin[k][:n].copy_(src)                        # host: the reference's .float() (patch_VAE.py:419)
                src = in_pin[k][:n]
            with torch.cuda.stream(s_in):
                s_in.wait_event(ev_done[k])                     # x_dev[k] is no longer being read (no-op the first time)
                x_dev[k][:n].copy_(src, non_blocking=True)      # (pageable source: blocks this thread until it has left)
                ev_in[k].record(s_in)
            compute.wait_event(ev_in[k])
            x = x_dev[k][:n]
            if zscore_on_device:
                x = ops.zscore_patch(x)
            z_b, join = encode(x)
            z_a, _, _ = E.vq_forward(codebook, z_b, float(model.commitment_cost), want_scalars=False)
            if join is not None:
                join()                                          # the running-statistics replay ran beside the quantiser
            ev_done[k].record(compute)
            z_b, z_a = z_b.reshape(n, -1), z_a.reshape(n, -1)
            if res[0] is None:
                pinned = 4 * N * (z_b.shape[1] + z_a.shape[1]) <= pin_cap
                for q, zq in enumerate((z_b, z_a)):
                    if pinned:
                        try:
                            res[q] = torch.empty((N, zq.shape[1]), dtype=torch.float32, pin_memory=True)
                            continue
                        except RuntimeError:                    # the host refuses to pin that much: pageable results
                            pinned = False
                    res[q] = torch.empty((N, zq.shape[1]), dtype=torch.float32).fill_(0)
            z_b.record_stream(s_out)
            z_a.record_stream(s_out)
            sent[k] = helper.submit(hand_back, lo, n, z_b, z_a, ev_done[k])
        for f in sent:
            if f is not None:
                f.result()
        s_out.synchronize()
    return res[0].numpy(), res[1].numpy()


def encode_patches_sharded(model, patches, device="cuda:0", batch_size=1024, zscore_on_device=False, group=None, dst=0):
    """encode_patches over a torch.distributed group: patches are independent (per-sample BatchNorm statistics), so rank r
    encodes the contiguous shard dist.shard_range(N, r, world) with no collective on the data path, and the (N, D*h*w)
    results are handed over on the host to rank `dst` (the one that writes the pickles) in rank order = input order
    (patch_VAE.py:454,459 stack in file-path order); the other ranks return None.
    BatchNorm running statistics advance per rank by that rank's shard only (they are not part of the outputs)."""
    from . import dist as D
    import torch.distributed as tdist
    patches = torch.as_tensor(patches)
    world = D.world_size(group)
    rank = tdist.get_rank(group) if world > 1 else 0
    lo, hi = D.shard_range(patches.shape[0], rank, world)
    z_b, z_a = encode_patches(model, patches[lo:hi], device=device, batch_size=batch_size, zscore_on_device=zscore_on_device)
    return D.gather_shards((z_b, z_a), group=group, dst=dst) if world > 1 else (z_b, z_a)


def process_VAE(raw_folder, supp_folder, sites, config_, gpu=0, network_module=None, **kwargs):
    """Same contract as the reference: reads <raw>/<well>_file_paths.pkl and <well>_static_patches.pkl,
    loads <weights>/model.pt, writes <raw>/<model_name>/<well>_latent_space[_after].pkl (protocol 4)."""
    le = config_.latent_encoding
    channels = le.channels
    network = le.network
    weights_dir = le.weights
    assert len(channels) > 0, "At least one channel must be specified"
    model_path = os.path.join(weights_dir, 'model.pt')
    model_name = os.path.basename(weights_dir)
    output_dir = os.path.join(raw_folder, model_name)
    os.makedirs(output_dir, exist_ok=True)
    assert len(set(site[:2] for site in sites)) == 1, "Sites should be from a single well/condition"
    well = sites[0][:2]

    with open(os.path.join(raw_folder, '%s_file_paths.pkl' % well), 'rb') as f:
        fs = pickle.load(f)
    with open(os.path.join(raw_folder, '%s_static_patches.pkl' % well), 'rb') as f:
        dataset = pickle.load(f)
    on_dev = bool(kwargs.get("zscore_on_device", False))
    if on_dev:
        dataset = torch.from_numpy(np.ascontiguousarray(np.squeeze(dataset)))       # raw float64; z-scored per batch on the GPU
    else:
        dataset = torch.from_numpy(zscore_patch(np.squeeze(dataset)))
        try:                                        # the .float() of patch_VAE.py:419, into pinned memory: encode_patches
            dataset = torch.empty(dataset.shape, dtype=torch.float32, pin_memory=True).copy_(dataset)   # then runs at the PCIe rate
        except RuntimeError:
            dataset = dataset.float()
    assert dataset.dim() == 4, "dataset tensor dimension can only be 4, not {}".format(dataset.dim())
    assert len(fs) == dataset.shape[0]
    device = torch.device('cuda:%d' % gpu)
    if 'VAE' not in network:
        raise ValueError('Network {} is not available'.format(network))
    if network_module is None:
        from . import vq_vae as network_module
    model = getattr(network_module, network)(num_inputs=dataset.shape[1],
                                             num_hiddens=le.num_hiddens,
                                             num_residual_hiddens=le.num_residual_hiddens,
                                             num_residual_layers=2,
                                             num_embeddings=le.num_embeddings,
                                             gpu=True).to(device)
    try:
        model.load_state_dict(torch.load(model_path, map_location=device))
    except Exception as ex:
        print(ex)
        raise ValueError("Error in loading model weights for VQ-VAE")
    z_b, z_a = encode_patches(model, dataset, device=device, batch_size=kwargs.get("batch_size", 1024),
                              zscore_on_device=on_dev)
    for name, dats in (('%s_latent_space.pkl' % well, z_b), ('%s_latent_space_after.pkl' % well, z_a)):
        with open(os.path.join(output_dir, name), 'wb') as f:
            pickle.dump(dats, f, protocol=4)
    if getattr(le, "save_output", False):
        save_recon_samples(model, dataset, output_dir, device, zscored=not on_dev)
    return z_b, z_a


def save_recon_samples(model, dataset, output_dir, device, zscored=True, n_samples=20):
    """patch_VAE.py:464-489 (`save_output`): 20 samples drawn with np.random.seed(0), each reconstructed by a batch-of-one
    `model(sample)[0]` (train mode, like every call of this path) on the HIP pipeline.  The arrays go to
    <output_dir>/recon_<i>.npz (sample, output); the 2 x 2 figure recon_<i>.jpg of the reference is drawn from them when
    matplotlib is importable (plotting itself is outside the hot path: same layout, a percentile stretch for contrast)."""
    np.random.seed(0)
    random_inds = np.random.randint(0, len(dataset), (n_samples,))
    written = []
    for i in random_inds:
        sample = dataset[i:(i + 1)]
        if not zscored:                                        # raw float64 patches were handed over: z-score this one
            sample = torch.from_numpy(zscore_patch(sample.numpy()))
        sample = sample.float().to(device)
        with torch.no_grad():
            output = model(sample)[0]
        a, b = sample[0].cpu().numpy(), output[0].detach().cpu().numpy()
        path = os.path.join(output_dir, 'recon_%d.npz' % i)
        np.savez(path, sample=a, output=b)
        written.append(path)
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except Exception:
            continue

        def stretch(im, tol=1):                                # contrast limits at the tol / 100 - tol percentiles
            lo, hi = np.percentile(im, [tol, 100 - tol])
            return np.clip((im - lo) / max(hi - lo, 1e-12), 0, 1)
        fig, ax = plt.subplots(2, 2, squeeze=False)
        fig.set_size_inches((15, 10))
        ims = [a[0], b[0], a[min(1, a.shape[0] - 1)], b[min(1, b.shape[0] - 1)]]
        for axis, im, name in zip(ax.flatten(), ims, ['phase', 'phase_recon', 'im_retard', 'retard_recon']):
            axis.imshow(stretch(im), cmap='gray')
            axis.axis('off')
            axis.set_title(name, fontsize=12)
        fig.savefig(os.path.join(output_dir, 'recon_%d.jpg' % i), dpi=100, bbox_inches='tight')
        plt.close(fig)
    return written
