#!/usr/bin/env python3
"""Parallel-imaging reconstruction driver: non-Cartesian SENSE by conjugate gradients.

    python -m indigo_amd.pics [-i ITER] [--lamda L] [-O LEVEL] [--crop "COIL:2,TIME:1"] [--no-fuse] scan.h5 | scan.npz

The counterpart of the reference's driver script (examples/pics.py:20-95 arguments, data layout and tree
construction, :179-233 recipe, normal equations, CG, output): reads `data` (k-space), `maps` (coil sensitivities)
and `traj` (trajectory, in units of 1/FOV pixels) with the reference's BART-style dimension order

    READ, PHS1, PHS2, COIL, MAPS, ..., TIME (dimension 10)            (stored reversed, hence the `.T`)

builds  A = KronI(C, NUFFT) * VStack(Diag(maps_c)),  rewrites it with `sense_recipe(level)` (= pics.py -O<level>),
and -- where the backend has zero-pad-aware transforms for the grid -- `FuseZpadFFT`, solves
(A^H A + lamda I) x = A^H y with `Backend.cg`, and writes the image back as `rec`.

Containers: HDF5 (`.h5`, the reference's format; needs h5py, which this image does not ship) or NumPy `.npz` with
the same three arrays in the same orientation; the result goes back into the HDF5 file as dataset `rec`, or next to
an `.npz` input as `<name>.rec.npy`.

The backend is the MI355X one (`hip`); there is no CPU fallback in the product.  `main(argv, backend=...)` lets the
CPU test-suite drive the same code with the numpy oracle backend.
"""
import argparse
import logging
import os
import sys

import numpy as np

log = logging.getLogger("pics")


class dim:
    READ, PHS1, PHS2, COIL, MAPS, TIME, NDIM = 0, 1, 2, 3, 4, 10, 20


def parse(argv):
    ap = argparse.ArgumentParser(prog="indigo_amd.pics", description="Parallel Imaging and Compressed Sensing (non-Cartesian SENSE, CG).")
    ap.add_argument('-i', type=int, default=20, help='number of CG iterations')
    ap.add_argument('--backend', type=str, default='hip', choices=['hip'])
    ap.add_argument('--device', type=int, default=0)
    ap.add_argument('--debug', type=int, default=logging.INFO, help='logging level')
    ap.add_argument('--crop', help='crop data before recon: --crop "COIL:2,TIME:4"')
    ap.add_argument('--lamda', type=float, default=0, help='Tikhonov regularisation parameter')
    ap.add_argument('-O', '--recipe', type=int, default=3, choices=range(5), help='optimization level (pics.py -O)')
    ap.add_argument('--osf', type=float, default=640 / 480, help='gridding oversampling factor (pics.py: 640/480)')
    ap.add_argument('--width', type=int, default=3, help='Kaiser-Bessel kernel half-width (Backend.NUFFT default)')
    ap.add_argument('--no-fuse', action='store_true', help='keep the -O tree as it is (no FuseZpadFFT)')
    ap.add_argument('data', nargs='?', default="scan.h5", help='k-space data: HDF5 (data/maps/traj) or .npz')
    return ap.parse_args(argv)


def load(path):
    """-> (data, maps, traj, writer): arrays as stored (reversed dimension order), writer(img_T) stores `rec`"""
    if path.endswith(".npz"):
        z = np.load(path)
        out = os.path.splitext(path)[0] + ".rec.npy"
        return z['data'], z['maps'], z['traj'], lambda rec: np.save(out, rec)
    try:
        import h5py
    except ImportError:
        raise SystemExit("pics: reading %s needs h5py; convert the scan to .npz (arrays data, maps, traj) or install h5py" % path)
    hdf = h5py.File(path, 'r+')

    def write(rec):
        if 'rec' in hdf:
            del hdf['rec']
        hdf.create_dataset('rec', data=rec)
        hdf.close()
    return hdf['data'][:], hdf['maps'][:], hdf['traj'][:], write


def crop_limits(spec):
    crops = [10 ** 6] * dim.NDIM
    if spec:
        names = {k: v for k, v in vars(dim).items() if k.isupper()}
        for item in spec.split(","):
            name, size = item.split(":")
            d = names[name.strip()]
            crops[-(d + 1)] = int(size)
            log.info("cropping dim %d to length %d", d, int(size))
    return crops


def reconstruct(B, ksp, mps, traj, iters=20, lamda=0.0, level=3, osf=640 / 480, width=3, fuse=True):
    """ksp: (1, readout, views, C, 1, ...), mps: (X, Y, Z, C, 1), traj: (3, readout, views) in pixels -> image (X, Y, Z, 1, ...)"""
    from indigo_amd.transforms import FuseZpadFFT, Optimize, sense_recipe
    from indigo_amd.transforms import reserve_for
    ksp = np.asarray(ksp, dtype=np.complex64)
    mps = np.asarray(mps, dtype=np.complex64)
    traj = np.array(traj, dtype=np.float64)
    ksp_nc_dims = ksp.shape
    img_dims = mps.shape[:3] + (1,) + ksp.shape[4:]
    log.info('img %s %s', img_dims, ksp.dtype)
    log.info('mps %s, ksp %s, trj %s', mps.shape, ksp.shape, traj.shape)
    for i in range(3):                                   # trajectory in units of the field of view (pics.py:69-72)
        traj[i] /= mps.shape[i]
    C = ksp.shape[dim.COIL]
    assert (ksp.shape[dim.TIME] if ksp.ndim > dim.TIME else 1) == 1, "No support for multiple timepoints."
    assert (mps.shape[dim.MAPS] if mps.ndim > dim.MAPS else 1) == 1, "No support for multiple maps."
    trj3 = traj.reshape(traj.shape[:3])
    F1 = B.NUFFT(ksp_nc_dims[:3], mps.shape[:3], trj3, width=width, oversamp=(osf, osf, osf), dtype=ksp.dtype)
    F = B.KronI(C, F1)
    S = B.VStack([B.Diag(mps[:, :, :, c].reshape(mps.shape[:3] + (1,))) for c in range(C)], name='maps')
    A = F * S
    A._name = 'SENSE1'
    recipe = sense_recipe(level)
    if fuse and level >= 3:
        recipe = recipe + [FuseZpadFFT]
    A = Optimize(recipe).visit(A)
    AHA = (A.H * A) + lamda * B.Eye(A.shape[1])
    AHA._name = 'SENSE'
    reserve_for(AHA, 1, slack_products=6)
    log.info("tree:\n%s", AHA.dump())
    log.info('using %d MB of device memory', (AHA.memusage() + 4 * AHA.shape[1] * ksp.dtype.itemsize) / 1e6)
    y = np.asfortranarray(ksp.reshape((-1, 1), order='F'))
    AHy = A.H * y
    AHy /= abs(AHy).max()
    x = np.zeros((AHA.shape[1], 1), dtype=ksp.dtype, order='F')
    hist = B.cg(AHA, AHy, x, maxiter=iters)
    log.info("residuals: %s", " ".join("%.3e" % h for h in (hist or [])))
    return x.reshape(img_dims, order='F')


def main(argv=None, backend=None):
    args = parse(sys.argv[1:] if argv is None else argv)
    logging.basicConfig(level=args.debug)
    if backend is None:
        from indigo_amd.backends import get_backend
        backend = get_backend(args.backend, device_id=args.device)
    log.info("using backend: %s", type(backend).__name__)
    data, maps, traj, write = load(args.data)
    crops = crop_limits(args.crop)
    ksp = data[tuple(slice(0, min(n, c)) for n, c in zip(data.shape, crops[-data.ndim:]))].T
    mps = maps[tuple(slice(0, min(n, c)) for n, c in zip(maps.shape, crops[-maps.ndim:]))].T
    trj = traj[tuple(slice(0, min(n, c)) for n, c in zip(traj.shape, crops[-traj.ndim:]))].T
    img = reconstruct(backend, ksp, mps, trj, iters=args.i, lamda=args.lamda, level=args.recipe, osf=args.osf,
                      width=args.width, fuse=not args.no_fuse)
    write(img.T)
    log.info("reconstruction complete")
    return img


if __name__ == "__main__":
    main()
