"""TensorBoard event files without TensorBoard (tfevents.py) and the Monitor-style episode statistics that feed them.

Reference consumer: learned_controllers/visualize/learning_curves.py:35-121 (EventAccumulator over events.out.tfevents.*,
tags rollout/ep_rew_mean ... train/explained_variance).  TensorBoard is absent: the wire format is checked against the
protobuf runtime with descriptors built here from the published event.proto / summary.proto field numbers, and the CRC
against CRC-32C's standard check value -- parity with TensorBoard's own reader is unpinned.
"""
import os
import struct

import numpy as np
import pytest
import torch

from hcrl_amd import tfevents


def test_crc32c_check_value_and_mask():
    assert tfevents.crc32c(b"123456789") == 0xE3069283                     # the CRC-32C (Castagnoli) check value
    assert tfevents.crc32c(b"") == 0
    c = tfevents.crc32c(b"abc")
    assert tfevents.masked_crc32c(b"abc") == ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _event_message_class():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="hcrl_test_event.proto", package="hcrl_test", syntax="proto3")
    F = descriptor_pb2.FieldDescriptorProto
    val = fd.message_type.add(name="Value")
    val.field.add(name="tag", number=1, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    val.field.add(name="simple_value", number=2, type=F.TYPE_FLOAT, label=F.LABEL_OPTIONAL)
    summ = fd.message_type.add(name="Summary")
    summ.field.add(name="value", number=1, type=F.TYPE_MESSAGE, label=F.LABEL_REPEATED, type_name=".hcrl_test.Value")
    ev = fd.message_type.add(name="Event")
    ev.field.add(name="wall_time", number=1, type=F.TYPE_DOUBLE, label=F.LABEL_OPTIONAL)
    ev.field.add(name="step", number=2, type=F.TYPE_INT64, label=F.LABEL_OPTIONAL)
    ev.field.add(name="file_version", number=3, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    ev.field.add(name="summary", number=5, type=F.TYPE_MESSAGE, label=F.LABEL_OPTIONAL, type_name=".hcrl_test.Summary")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("hcrl_test.Event"))


def test_event_encoding_parses_with_the_protobuf_runtime():
    Event = _event_message_class()
    raw = tfevents.encode_event(1234.5, 300000000000, {"rollout/ep_rew_mean": -12.25, "train/approx_kl": 0.0125})
    ev = Event.FromString(raw)
    assert ev.wall_time == 1234.5 and ev.step == 300000000000
    got = {v.tag: v.simple_value for v in ev.summary.value}
    assert got == {"rollout/ep_rew_mean": -12.25, "train/approx_kl": struct.unpack("<f", struct.pack("<f", 0.0125))[0]}
    first = Event.FromString(tfevents.encode_event(1.0, 0, file_version="brain.Event:2"))
    assert first.file_version == "brain.Event:2" and len(first.summary.value) == 0
    # and the other way: bytes serialised by the runtime decode with our reader's field walker
    ev2 = Event(wall_time=2.0, step=7)
    ev2.summary.value.add(tag="x", simple_value=3.5)
    assert dict((n, v) for n, _w, v in tfevents._fields(ev2.SerializeToString()))[2] == 7


def test_writer_reader_round_trip_and_crc_detection(tmp_path):
    w = tfevents.EventFileWriter(str(tmp_path / "tb"))
    assert os.path.basename(w.path).startswith("events.out.tfevents.")
    for k in range(5):
        w.add_scalars({"rollout/ep_rew_mean": -100.0 + 10 * k, "train/value_loss": 1.0 / (k + 1)}, step=1000 * (k + 1), wall_time=50.0 + k)
    w.add_scalar("eval/mean_reward", 42.0, 5000)
    w.close()
    evs = tfevents.read_events(w.path)
    assert evs[0]["file_version"] == "brain.Event:2" and len(evs) == 7
    assert [e["step"] for e in evs[1:6]] == [1000, 2000, 3000, 4000, 5000] and evs[3]["wall_time"] == 52.0
    table = tfevents.load_scalars(str(tmp_path))
    assert [r[0] for r in table["rollout/ep_rew_mean"]] == [1000, 2000, 3000, 4000, 5000]
    assert [r[1] for r in table["rollout/ep_rew_mean"]] == [-100.0, -90.0, -80.0, -70.0, -60.0]
    assert table["eval/mean_reward"] == [(5000, 42.0, table["eval/mean_reward"][0][2])]
    raw = bytearray(open(w.path, "rb").read())
    raw[40] ^= 0x01                                                         # flip one payload bit
    bad = tmp_path / "events.out.tfevents.corrupt"
    bad.write_bytes(bytes(raw))
    with pytest.raises(ValueError, match="CRC mismatch"):
        tfevents.read_events(str(bad))


def _monitor_loop(rew, starts, final_start, run_ret, run_len):
    """What a Monitor wrapper per env would record, step by step."""
    T, N = rew.shape
    rets, lens = [], []
    run_ret, run_len = run_ret.copy(), run_len.copy()
    for t in range(T):
        done = starts[t + 1] if t + 1 < T else final_start
        for n in range(N):
            run_ret[n] += rew[t, n]; run_len[n] += 1
            if done[n] > 0:
                rets.append(run_ret[n]); lens.append(run_len[n]); run_ret[n] = 0.0; run_len[n] = 0
    return rets, lens, run_ret, run_len


@pytest.mark.parametrize("p_done", [0.0, 0.05, 0.5, 1.0])
def test_episode_stats_match_a_monitor_loop(p_done):
    from hcrl_amd.ppo import episode_stats_from_rollout
    rs = np.random.RandomState(int(p_done * 100))
    T, N = 16, 37
    run_ret, run_len = rs.normal(0, 5, N), rs.randint(0, 40, N).astype(np.float64)
    for _rollout in range(3):                                                # carries flow from one rollout to the next
        rew = rs.normal(0, 1, (T, N)).astype(np.float32)
        starts = (rs.rand(T, N) < p_done).astype(np.float32)
        final = (rs.rand(N) < p_done).astype(np.float32)
        sums, new_ret, new_len = episode_stats_from_rollout(torch.as_tensor(rew), torch.as_tensor(starts), torch.as_tensor(final),
                                                            torch.as_tensor(run_ret), torch.as_tensor(run_len))
        rets, lens, want_ret, want_len = _monitor_loop(rew.astype(np.float64), starts, final, run_ret, run_len)
        ret_sum, len_sum, count = sums.tolist()
        assert count == len(rets)
        assert ret_sum == pytest.approx(float(np.sum(rets)), abs=1e-9) and len_sum == float(np.sum(lens))
        assert np.allclose(new_ret.numpy(), want_ret, atol=1e-9) and np.array_equal(new_len.numpy(), want_len)
        run_ret, run_len = want_ret, want_len
