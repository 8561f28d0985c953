"""Generate tests/golden/ground_plane.npz by RUNNING the reference's two offline ground-embedding scripts
(/root/reference/tools/preprocess_data_kitti.py and preprocess_data_ddad.py, unmodified, via runpy) on a toy
calibration tree built here.  Build-container only (the reference never travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ground.py

Stand-ins the scripts need (neither library exists in this image): ``cv2.imread`` (PIL-backed: BGR uint8, or the raw
16-bit array with flag -1), ``IPython.embed`` (no-op), and for DDAD ``dgp.datasets.SynchronizedSceneDataset`` (a toy
dataset object that returns the calibration below in dgp's item layout).  All arithmetic — matrix products, inverse,
ray/plane division, arctan / rad2deg / around / int64 truncation, clipping, 255 fill — is the reference's own code.

The fixture stores the INPUTS (calibration matrices, image sizes, sparse depth maps) and the reference's OUTPUTS
(pe float64 maps, slope-class maps); tests compare oracle.ground_plane / slope_class (CPU) and ge_ground_plane /
ge_slope_class (MI355X) with them bit for bit."""
import glob
import os
import runpy
import shutil
import sys
import tempfile
import types

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
sys.dont_write_bytecode = True


def install_standins(ddad_samples):
    cv2 = types.ModuleType('cv2')

    def imread(path, flag=1):
        im = Image.open(path)
        a = np.asarray(im)
        if flag == -1:
            return a
        return np.ascontiguousarray(np.asarray(im.convert('RGB'))[:, :, ::-1])
    cv2.imread = imread
    sys.modules['cv2'] = cv2
    ipy = types.ModuleType('IPython')
    ipy.embed = lambda *a, **k: None
    sys.modules['IPython'] = ipy

    class _Mat:
        def __init__(self, m):
            self.matrix = np.asarray(m, dtype=np.float64)

    class _Datum:
        class datum:
            class image:
                filename = 'toy.png'

    class SynchronizedSceneDataset:
        """dgp item layout used by preprocess_data_ddad.py:20-33: dataset[i][0][cam] -> dict(rgb, intrinsics, extrinsics,
        pose); index 4 is the lidar datum."""

        def __init__(self, *a, **k):
            self.dataset_metadata = types.SimpleNamespace(directory='.')
            self.dataset_item_index = [(0, 0, [0, 1, 2, 3, 4])]

        def get_datum(self, *a):
            return _Datum()

        def __getitem__(self, i):
            cams = [dict(rgb=Image.fromarray(np.zeros((s['H'], s['W'], 3), np.uint8)), intrinsics=s['K'],
                         extrinsics=_Mat(s['extrinsics']), pose=_Mat(s['pose'])) for s in ddad_samples['cams']]
            return [cams + [dict(pose=_Mat(ddad_samples['lidar_pose']))]]
    dgp = types.ModuleType('dgp')
    dgp_ds = types.ModuleType('dgp.datasets')
    dgp_ds.SynchronizedSceneDataset = SynchronizedSceneDataset
    dgp.datasets = dgp_ds
    sys.modules['dgp'] = dgp
    sys.modules['dgp.datasets'] = dgp_ds


def rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def fmt(v):
    return ' '.join(f'{x:.6e}' for x in np.asarray(v).reshape(-1))


def sparse_depth(rng, H, W, frac=0.3):
    return np.where(rng.random((H, W)) < frac, rng.uniform(1.0, 80.0, (H, W)), 0.0)


def make_kitti_tree(root, rng):
    """Two calibration days (quarter-size images: the 2011_09_26 intrinsics of datasets/kitti.py:182-184 scaled by 1/4, the
    second day with a small camera pitch / roll), one drive each, two frames each."""
    out = {}
    H, W = 94, 311
    lines = []
    for di, (date, ang) in enumerate((('2011_09_26', (0.0, 0.0, 0.0)), ('2011_10_03', (0.012, -0.004, 0.007)))):
        P2 = np.array([[180.384425, 0.0, 152.389825, 11.21432], [0.0, 180.384425, 43.2135, 0.05409], [0.0, 0.0, 1.0, 2.745884e-03]])
        R0 = rot(*ang)
        velo_R = np.array([[7.533745e-03, -9.999714e-01, -6.166020e-04], [1.480249e-02, 7.280733e-04, -9.998902e-01],
                           [9.998621e-01, 7.523790e-03, 1.480755e-02]])
        velo_T = np.array([-4.069766e-03, -7.631618e-02, -2.717806e-01])
        droot = os.path.join(root, 'data/kitti/input', date)
        os.makedirs(droot)
        cam = ['calib_time: 09-Jan-2012 13:57:47', 'corner_dist: 9.950000e-02']
        for c in range(4):
            cam += [f'S_0{c}: 1.392000e+03 5.120000e+02', f'K_0{c}: ' + fmt(np.eye(3)), f'D_0{c}: ' + fmt(np.zeros(5)),
                    f'R_0{c}: ' + fmt(np.eye(3)), f'T_0{c}: ' + fmt(np.zeros(3)), f'S_rect_0{c}: 1.242000e+03 3.750000e+02',
                    f'R_rect_0{c}: ' + fmt(R0 if c == 0 else np.eye(3)), f'P_rect_0{c}: ' + fmt(P2 if c == 2 else np.eye(3, 4))]
        open(os.path.join(droot, 'calib_cam_to_cam.txt'), 'w').write('\n'.join(cam) + '\n')
        open(os.path.join(droot, 'calib_velo_to_cam.txt'), 'w').write(
            '\n'.join(['calib_time: 15-Mar-2012 11:37:16', 'R: ' + fmt(velo_R), 'T: ' + fmt(velo_T)]) + '\n')
        drive = f'{date}_drive_0001_sync'
        img_dir = os.path.join(droot, drive, 'image_02/data')
        gt_dir = os.path.join(root, 'data/kitti/gt_depth', drive, 'proj_depth/groundtruth/image_02')
        os.makedirs(img_dir)
        os.makedirs(gt_dir)
        Image.fromarray(np.zeros((H, W, 3), np.uint8)).save(os.path.join(img_dir, '0000000000.png'))
        # the file values the reference parses (6 significant digits), not the arrays above
        rd = lambda s: np.array([float(x) for x in s.split(' ')[1:]])
        out[f'kitti{di}_P2'] = rd(cam[25]).reshape(3, 4)
        out[f'kitti{di}_R0'] = rd(cam[8]).reshape(3, 3)
        Tr = np.eye(4)
        Tr[:3, :3] = rd('R: ' + fmt(velo_R)).reshape(3, 3)
        Tr[:3, 3] = rd('T: ' + fmt(velo_T))
        out[f'kitti{di}_Tr'] = Tr
        for f in range(2):
            name = f'{f:010d}.png'
            gt16 = (sparse_depth(rng, H, W) * 256).astype(np.uint16)
            Image.fromarray(gt16).save(os.path.join(gt_dir, name))
            out[f'kitti{di}_gt16_{f}'] = gt16
            lines.append(f'{date}/{drive}/image_02/data/{name} {drive}/proj_depth/groundtruth/image_02/{name} 721.5377')
    lines.append('2011_09_26/2011_09_26_drive_0001_sync/image_02/data/0000000099.png None 721.5377')
    open(os.path.join(root, 'data/kitti/kitti_eigen_train.txt'), 'w').write('\n'.join(lines) + '\n')
    return out


def make_ddad(root, rng):
    """Four cameras (eighth-size 152 x 242, intrinsics ~ DDAD's 2181 px focal / 8) with distinct poses + a lidar pose."""
    H, W = 152, 242
    cams = []
    for i, (yaw, pitch) in enumerate(((0.0, 0.004), (-1.02, -0.006), (1.03, 0.003), (3.13, 0.008))):
        K = np.array([[272.6 + 3 * i, 0.0, 120.8 - i], [0.0, 272.6 + 3 * i, 75.1 + 2 * i], [0.0, 0.0, 1.0]])
        # camera frame (x right, y down, z forward) -> world (x forward, y left, z up), then yaw / pitch
        base = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])
        pose = np.eye(4)
        pose[:3, :3] = rot(0.0, pitch, yaw) @ base
        pose[:3, 3] = [1.5 + 0.1 * i, 0.2 * (i - 1.5), 1.55 + 0.01 * i]
        cams.append(dict(H=H, W=W, K=K, pose=pose, extrinsics=np.eye(4)))
    lidar_pose = np.eye(4)
    lidar_pose[:3, :3] = rot(0.002, -0.003, 0.01)
    lidar_pose[:3, 3] = [1.2, 0.0, 1.9]
    names = ['CAMERA_01', 'CAMERA_05', 'CAMERA_06', 'CAMERA_09']
    out, lines = {}, []
    for i, n in enumerate(names):
        os.makedirs(os.path.join(root, 'data/DDAD/pe_public_debug', n))
        d = os.path.join(root, 'data/DDAD/depth', n)
        os.makedirs(d)
        gt = sparse_depth(rng, H, W, 0.2).astype(np.float32)
        np.savez(os.path.join(d, '000.npz'), depth=gt)
        out[f'ddad{i}_gt'] = gt
        out[f'ddad{i}_K'] = cams[i]['K']
        out[f'ddad{i}_pose'] = cams[i]['pose']
        lines.append(f'rgb/{n}/000.png data/DDAD/depth_val/{n}/000.npz')
    out['ddad_lidar_pose'] = lidar_pose
    os.makedirs(os.path.join(root, 'splits'))
    open(os.path.join(root, 'splits/ddad_train_split.txt'), 'w').write('\n'.join(lines) + '\n')
    return dict(cams=cams, lidar_pose=lidar_pose), out


def main():
    rng = np.random.default_rng(2024)
    root = tempfile.mkdtemp(prefix='ge_ground_')
    cwd = os.getcwd()
    try:
        arrays = make_kitti_tree(root, rng)
        ddad_samples, ddad_arrays = make_ddad(root, rng)
        arrays.update(ddad_arrays)
        install_standins(ddad_samples)
        os.chdir(root)
        print('running the reference tools/preprocess_data_kitti.py')
        runpy.run_path(os.path.join(REF, 'tools/preprocess_data_kitti.py'), run_name='__main__')
        print('running the reference tools/preprocess_data_ddad.py')
        runpy.run_path(os.path.join(REF, 'tools/preprocess_data_ddad.py'), run_name='__main__')
        for di, date in enumerate(('2011_09_26', '2011_10_03')):
            arrays[f'kitti{di}_pe'] = np.asarray(np.load(f'data/kitti/input/{date}/pe/pe_165.npy'))
            ks = sorted(glob.glob(f'data/kitti/slope_range_5_5_interval_1/{date}_drive_0001_sync/proj_depth/groundtruth/image_02/*.npz'))
            assert len(ks) == 2, ks
            for f, p in enumerate(ks):
                arrays[f'kitti{di}_k_{f}'] = np.load(p)['k_img']
        for i, n in enumerate(['CAMERA_01', 'CAMERA_05', 'CAMERA_06', 'CAMERA_09']):
            arrays[f'ddad{i}_pe'] = np.asarray(np.load(f'data/DDAD/pe_public_debug/{n}/ddad_pe.npz')['pe'])
            arrays[f'ddad{i}_k'] = np.load(f'data/DDAD/depth/{n}/000_slope_public_debug.npz')['k_img']
    finally:
        os.chdir(cwd)
        shutil.rmtree(root, ignore_errors=True)
    for k, v in sorted(arrays.items()):
        print(f'  {k:18s} {str(v.dtype):8s} {v.shape}')
    path = os.path.join(HERE, 'ground_plane.npz')
    np.savez_compressed(path, **arrays)
    print(f'wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)')


if __name__ == '__main__':
    main()
