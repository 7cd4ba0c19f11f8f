"""runtime.SamplerAheadPipeline: two graphs per batch (SA1's sampling | the rest) on sampler / dense streams give the bits of the
eager forward for every submitted batch, also when the sampler streams run slots ahead and slots are reused."""
import numpy as np
import pytest

from conftest import s_scene


def _small_hp(pn2, b, n):
    hp = dict(pn2.model.SEMANTIC_HYPERPARAMS)
    hp.update(batch_size=b, num_point=n, l1_npoint=256, l2_npoint=64, l3_npoint=16, l4_npoint=4,
              l1_nsample=16, l2_nsample=16, l3_nsample=8, l4_nsample=4)
    return hp


def _cloud(seed, b, n, dev):
    import torch
    rs = np.random.RandomState(seed)
    pc = np.concatenate([s_scene(seed, b, n)[..., :3], rs.random_sample((b, n, 3)).astype(np.float32)], axis=2)
    return torch.from_numpy(pc).to(dev)


@pytest.mark.gpu
@pytest.mark.parametrize("slots,ns,nd", [(1, 1, 1), (3, 2, 1), (4, 2, 2)])
def test_sampler_ahead_pipeline_equals_the_eager_forward(pn2, cuda, slots, ns, nd):
    import torch
    tfu = pn2.util.tf_util
    B, N = 2, 2048
    hp = _small_hp(pn2, B, N)
    tfu.set_default_store(tfu.VariableStore(device=cuda, seed=3))
    batches = [_cloud(100 + i, B, N, cuda) for i in range(slots)]
    with torch.no_grad():
        pn2.model.get_sa_fp_features(batches[0], False, hp)  # creates the variables
    pipe = pn2.runtime.SamplerAheadPipeline(lambda x: pn2.model.sa1_samples(x, hp),
                                            lambda x, s: pn2.model.get_sa_fp_features(x, False, hp, sa1=s)[0],
                                            batches, sampler_streams=ns, dense_streams=nd)
    # resident batches
    for k in range(slots):
        y = pipe.step()
        torch.cuda.synchronize()
        with torch.no_grad():
            ref = pn2.model.get_sa_fp_features(batches[k], False, hp)[0]
        assert torch.equal(y, ref), "slot %d" % k
    # fresh inputs, submitted back to back (the sampler streams run ahead, every slot is reused), outputs collected per step
    fresh = [_cloud(200 + i, B, N, cuda) for i in range(3 * slots + 1)]
    outs, pending = [None] * len(fresh), {}
    for i, x in enumerate(fresh):
        k = pipe.count % slots
        if k in pending:  # read a slot's output before the slot is submitted again
            j, yy = pending.pop(k)
            pipe.consumed[k].synchronize()
            outs[j] = yy.clone()
        pending[k] = (i, pipe.step(x))
    torch.cuda.synchronize()
    for j, yy in pending.values():
        outs[j] = yy.clone()
    with torch.no_grad():
        for i, x in enumerate(fresh):
            ref = pn2.model.get_sa_fp_features(x, False, hp)[0]
            assert torch.equal(outs[i], ref), "fresh batch %d" % i


@pytest.mark.gpu
def test_samples_given_ahead_equal_the_module_sampling_itself(pn2, cuda):
    import torch
    tfu = pn2.util.tf_util
    B, N = 2, 1024
    hp = _small_hp(pn2, B, N)
    tfu.set_default_store(tfu.VariableStore(device=cuda, seed=4))
    x = _cloud(7, B, N, cuda)
    with torch.no_grad():
        a, ea = pn2.model.get_sa_fp_features(x, False, hp)
        s = pn2.model.sa1_samples(x, hp)
        b, eb = pn2.model.get_sa_fp_features(x, False, hp, sa1=s)
    assert torch.equal(a, b)
    for u, v in zip(ea["xyzs"], eb["xyzs"]):
        assert torch.equal(u, v)
