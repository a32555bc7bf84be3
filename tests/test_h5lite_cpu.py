"""h5lite (dependency-free HDF5) against files a known-good writer produced -- h5py 3.3.0 / libhdf5 1.10.6, generated in
the build container by tools/gen_h5_fixtures.py and committed under tests/golden/h5/ -- in the layouts of the
reference (hdf5_layouts/Readme.md:105-117 input; test_ensemble.py:121-132, util.py:300-310 output), plus write -> read
round trips and the loud refusals.  CPU only."""
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN
import dfl_amd  # noqa: F401
from dfl_amd import dataset, h5lite

H5 = os.path.join(GOLDEN, 'h5')
NAMES = ['FH-l', 'FH-r', 'GSN-l', 'GSN-r', 'IOF-l', 'IOF-r', 'MOF-l', 'MOF-r', 'SPS-l', 'SPS-r', 'IPS-l', 'IPS-r',
         'ASIS-l', 'ASIS-r']


@pytest.fixture(scope='module')
def expected():
    return dict(np.load(os.path.join(H5, 'expected.npz')))


def test_reads_the_preprocessed_layout_written_by_h5py(expected):
    with h5lite.File(os.path.join(H5, 'preproc_default.h5'), 'r') as f:
        assert f.keys() == sorted(['01', '02', '11', 'land-names', 'many', 'edge', 'be16', 'fixed', 'scalar_f64', 'wide'])
        assert int(f['land-names/num-lands'][()]) == 14
        assert [f['land-names/land-%02d' % i][()].decode() for i in range(14)] == NAMES       # variable-length strings
        for pat in ('01', '02', '11'):
            for k in ('projs', 'segs', 'lands'):                     # contiguous f32 / gzip-9 u8 / shuffle+gzip f64
                d = f[pat + '/' + k]
                assert d.shape == expected[pat + '/' + k].shape and d.dtype == expected[pat + '/' + k].dtype
                assert np.array_equal(d[:], expected[pat + '/' + k])
                assert np.array_equal(d[()], expected[pat + '/' + k])
        assert f['01/projs'].shape == (3, 12, 10) and len(f['01/projs']) == 3         # dataset.py:331-337
        m = f['many']                                                # 150 chunks: a two-level chunk B-tree
        assert np.array_equal(m[:], expected['many'])
        assert np.array_equal(m[77], expected['many'][77]) and np.array_equal(m[-1], expected['many'][-1])
        assert np.array_equal(m[10:20, 1:3], expected['many'][10:20, 1:3])
        assert np.array_equal(m[140:, ..., 2], expected['many'][140:, ..., 2])
        assert int(m[3, 2, 1]) == int(expected['many'][3, 2, 1])
        assert np.array_equal(f['edge'][:], expected['edge'])        # chunks hanging over the edge
        assert np.array_equal(f['edge'][1:4, 2:9], expected['edge'][1:4, 2:9])
        assert np.array_equal(f['be16'][:], expected['be16']) and f['be16'].dtype == np.int16     # big-endian on disk
        assert f['fixed'][()] == b'abc' and float(f['scalar_f64'][()]) == 2.5
        w = f['wide']                                                # 40 links: several symbol-table nodes
        assert len(w) == 40 and w.keys()[0] == 'item-000' and all(int(w['item-%03d' % i][()]) == i * i for i in range(40))
        assert 'land-names/land-13' in f and 'land-names/land-14' not in f and '07' not in f
        with pytest.raises(KeyError):
            f['07/projs']
        with pytest.raises(IndexError):
            m[150]


def test_reads_latest_format_files(expected):
    """libver='latest': superblock 3, version-2 object headers, compact link messages, single-chunk index."""
    with h5lite.File(os.path.join(H5, 'preproc_latest.h5'), 'r') as f:
        assert f.keys() == ['04', 'land-names'] and f['04'].keys() == ['lands', 'projs', 'segs']
        assert f['land-names/land-01'][()] == b'GSN-r' and int(f['land-names/num-lands'][()]) == 2
        assert np.array_equal(f['04/projs'][:], expected['01/projs'])
        assert np.array_equal(f['04/segs'][:], expected['01/segs'])
        assert np.array_equal(f['04/lands'][1:], expected['01/lands'][1:])


def test_reads_the_output_layout_written_by_h5py(expected):
    with h5lite.File(os.path.join(H5, 'nn_out.h5'), 'r') as f:
        assert f['nn-segs'].dtype == np.uint8 and f['nn-heats'].dtype == np.float32
        assert np.array_equal(f['nn-segs'][:], expected['nn-segs']) and np.array_equal(f['nn-heats'][:], expected['nn-heats'])
        assert np.array_equal(f['nn-heats'][2, 1], expected['nn-heats'][2, 1])


def test_loader_helpers_read_hdf5_without_h5py():
    p = os.path.join(H5, 'preproc_default.h5')
    assert dataset.get_num_lands_from_dataset(p) == 14                # dataset.py:339-346
    assert dataset.get_land_names_from_dataset(p) == NAMES            # dataset.py:348-365
    get, close = dataset._open_container(p)
    assert np.asarray(get('02/segs')).shape == (2, 12, 10)
    close()


def test_write_read_round_trip(tmp_path):
    rng = np.random.RandomState(1)
    p = str(tmp_path / 'out.h5')
    segs = rng.randint(0, 7, size=(130, 9, 11)).astype(np.uint8)
    heats = rng.rand(130, 3, 9, 11).astype(np.float32)
    f = dataset.open_output_container(p)                              # test_ensemble.py:121-129
    g = f.create_group('land-names')
    g['num-lands'] = 3
    for i in range(3):
        g['land-%02d' % i] = NAMES[i]
    d = f.create_dataset('nn-segs', (130, 9, 11), dtype='u1', chunks=(1, 9, 11), compression='gzip', compression_opts=9)
    h = f.create_dataset('nn-heats', (130, 3, 9, 11), chunks=(1, 1, 9, 11), compression='gzip', compression_opts=9)
    for i in range(130):                                              # util.py:361-373: one image at a time
        d[i, :, :] = segs[i]
        h[i, :, :, :] = heats[i]
    e = f.create_dataset('edge', (5, 9), dtype='f8', chunks=(2, 4))   # partial chunks, assigned piecewise
    ed = rng.randn(5, 9)
    e[:3] = ed[:3]
    e[3:, :5] = ed[3:, :5]
    e[3:, 5:] = ed[3:, 5:]
    f['a/b/c'] = np.arange(5, dtype=np.int32)
    f['pi'] = 3.25
    f.flush()
    f.close()
    assert h5lite.is_hdf5(p)
    with h5lite.File(p, 'r') as r:
        assert r.keys() == ['a', 'edge', 'land-names', 'nn-heats', 'nn-segs', 'pi']
        assert np.array_equal(r['nn-segs'][:], segs) and r['nn-segs'].dtype == np.uint8
        assert np.array_equal(r['nn-heats'][:], heats) and r['nn-heats'].dtype == np.float32
        assert np.array_equal(r['nn-heats'][100, 2], heats[100, 2])
        assert np.array_equal(r['edge'][:], ed)
        assert np.array_equal(r['a/b/c'][:], np.arange(5)) and float(r['pi'][()]) == 3.25
    assert dataset.get_num_lands_from_dataset(p) == 3 and dataset.get_land_names_from_dataset(p) == NAMES[:3]
    # on-disk structure the reference's h5py call asks for: gzip level 9, one chunk per image / per heat map
    raw = open(p, 'rb').read()
    assert raw[:8] == b'\x89HDF\r\n\x1a\n' and struct.unpack('<Q', raw[40:48])[0] == len(raw)      # end-of-file address


def test_rewriting_a_streamed_chunk_is_refused(tmp_path):
    f = h5lite.File(str(tmp_path / 'x.h5'), 'w')
    d = f.create_dataset('d', (4, 3, 3), dtype='u1', chunks=(1, 3, 3), compression='gzip')
    d[0] = 1
    with pytest.raises(h5lite.H5Error):
        d[0] = 2
    with pytest.raises(h5lite.H5Error):
        f.create_dataset('d', (1,), dtype='u1')
    with pytest.raises(h5lite.H5Error):
        f.create_dataset('e', (4, 4), dtype='u1', compression='lzf')
    f.close()
    with h5lite.File(str(tmp_path / 'x.h5')) as r:
        assert r['d'][0].min() == 1 and r['d'][1:].max() == 0        # chunks never written read back as the fill value


def test_not_hdf5_and_unsupported_features_fail_loudly(tmp_path):
    p = tmp_path / 'junk.h5'
    p.write_bytes(b'not an hdf5 file at all' * 10)
    assert not h5lite.is_hdf5(str(p))
    with pytest.raises(h5lite.H5Error):
        h5lite.File(str(p), 'r')
    with pytest.raises(h5lite.H5Error):
        h5lite.File(str(p), 'a')
