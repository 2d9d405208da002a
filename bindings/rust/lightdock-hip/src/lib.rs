//! Thin, safe wrapper over `include/lightdock_hip.h` shaped like lightdock-rust's own types:
//! `HipScore` is what `DFIRE::new` / `DNA::new` / `PYDOCK::new` return (`Box<dyn Score>`,
//! lightdock-rust src/scoring.rs:11-19), `HipGso` is `GSO` (src/lib.rs:21-58) for a batch of swarms.
//!
//! The `Score` trait and `Quaternion` live in the lightdock crate; to keep this crate free of a
//! dependency cycle the trait impl is written against a local mirror and re-implemented in one
//! line inside lightdock-rust (Level 1).
use std::ffi::{CStr, CString};
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct LdMolecule {
    pub n_atoms: usize,
    pub coordinates: *const f64,
    pub dfire_types: *const u32,
    pub ele_charges: *const f64,
    pub vdw_charges: *const f64,
    pub vdw_radii: *const f64,
    pub n_membrane: usize,
    pub membrane: *const u32,
    pub n_restraint_groups: usize,
    pub restraint_offsets: *const u32,
    pub restraint_atoms: *const u32,
    pub num_anm: usize,
    pub nmodes: *const f64,
}

#[repr(C)]
pub struct LdScorerDesc {
    pub method: c_int, // 0 DFIRE, 1 DNA, 2 PYDOCK
    pub use_anm: c_int,
    pub receptor: LdMolecule,
    pub ligand: LdMolecule,
    pub potential: *const f64,
}

extern "C" {
    fn ld_init(device: c_int) -> c_int;
    fn ld_last_error() -> *const c_char;
    fn ld_scorer_create(desc: *const LdScorerDesc) -> *mut c_void;
    fn ld_scorer_destroy(s: *mut c_void);
    fn ld_scorer_pose_len(s: *const c_void) -> usize;
    fn ld_scorer_energy(s: *mut c_void, t: *const f64, q_wxyz: *const f64, rec_nm: *const f64, lig_nm: *const f64,
                        out: *mut f64) -> c_int;
    fn ld_scorer_energy_batch(s: *mut c_void, n: usize, poses: *const f64, stride: usize, out: *mut f64) -> c_int;
    fn ld_gso_create(s: *mut c_void, n_swarms: usize, n_glowworms: usize, positions: *const f64, seeds: *const u64) -> *mut c_void;
    fn ld_gso_destroy(g: *mut c_void);
    fn ld_gso_run(g: *mut c_void, steps: u32) -> c_int;
    fn ld_gso_save(g: *mut c_void, swarm: usize, step: u32, dir: *const c_char) -> c_int;
    fn ld_gso_save_many(g: *mut c_void, n: usize, swarms: *const usize, dirs: *const *const c_char, step: u32) -> c_int;
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(ld_last_error()).to_string_lossy().into_owned() }
}

/// The vectors of a `DFIREDockingModel` / `DNADockingModel` (src/dfire.rs:104-112, src/dna.rs:235-246),
/// with `active_restraints` flattened to CSR (group order is irrelevant, src/scoring.rs:21-36).
#[derive(Default)]
pub struct ModelArrays {
    pub coordinates: Vec<[f64; 3]>,
    pub dfire_types: Vec<u32>,
    pub ele_charges: Vec<f64>,
    pub vdw_charges: Vec<f64>,
    pub vdw_radii: Vec<f64>,
    pub membrane: Vec<u32>,
    pub restraint_offsets: Vec<u32>,
    pub restraint_atoms: Vec<u32>,
    pub num_anm: usize,
    pub nmodes: Vec<f64>,
}

impl ModelArrays {
    fn view(&self) -> LdMolecule {
        let opt = |v: &Vec<f64>| if v.is_empty() { std::ptr::null() } else { v.as_ptr() };
        LdMolecule {
            n_atoms: self.coordinates.len(),
            coordinates: self.coordinates.as_ptr() as *const f64,
            dfire_types: if self.dfire_types.is_empty() { std::ptr::null() } else { self.dfire_types.as_ptr() },
            ele_charges: opt(&self.ele_charges),
            vdw_charges: opt(&self.vdw_charges),
            vdw_radii: opt(&self.vdw_radii),
            n_membrane: self.membrane.len(),
            membrane: self.membrane.as_ptr(),
            n_restraint_groups: self.restraint_offsets.len().saturating_sub(1),
            restraint_offsets: self.restraint_offsets.as_ptr(),
            restraint_atoms: self.restraint_atoms.as_ptr(),
            num_anm: self.num_anm,
            nmodes: opt(&self.nmodes),
        }
    }
}

/// Thread-compatibility: `ld_scorer_energy` reuses the handle's device workspaces and its stream, so a
/// handle serves ONE thread at a time.  The raw pointer field makes `HipScore` neither `Send` nor
/// `Sync` (which the compiler then enforces): exactly what the reference needs, whose `Score` trait has
/// no `Send + Sync` bound and whose binary scores on one worker thread (src/bin/lightdock-rust.rs:79-85).
/// A multi-threaded host creates one `HipScore` per thread (one per GPU in practice).
pub struct HipScore {
    handle: *mut c_void,
}

impl HipScore {
    /// `method`: 0 DFIRE (needs `potential`, the 169*169*20 values of data/DCparams), 1 DNA, 2 PYDOCK.
    /// Panics where the reference's constructors panic.
    pub fn new(method: i32, receptor: &ModelArrays, ligand: &ModelArrays, use_anm: bool, potential: Option<&[f64]>) -> Self {
        let desc = LdScorerDesc {
            method,
            use_anm: use_anm as c_int,
            receptor: receptor.view(),
            ligand: ligand.view(),
            potential: potential.map_or(std::ptr::null(), |p| p.as_ptr()),
        };
        unsafe {
            if ld_init(-1) != 0 {
                panic!("{}", last_error());
            }
            let handle = ld_scorer_create(&desc);
            if handle.is_null() {
                panic!("{}", last_error());
            }
            HipScore { handle }
        }
    }

    pub fn pose_len(&self) -> usize {
        unsafe { ld_scorer_pose_len(self.handle) }
    }

    /// `Score::energy` (src/scoring.rs:11-19); `rotation` = (w, x, y, z).
    pub fn energy(&self, translation: &[f64], rotation: [f64; 4], rec_nmodes: &[f64], lig_nmodes: &[f64]) -> f64 {
        let ptr = |v: &[f64]| if v.is_empty() { std::ptr::null() } else { v.as_ptr() };
        let mut e = 0.0;
        let rc = unsafe { ld_scorer_energy(self.handle, translation.as_ptr(), rotation.as_ptr(), ptr(rec_nmodes), ptr(lig_nmodes), &mut e) };
        assert_eq!(rc, 0, "{}", last_error());
        e
    }

    /// All glowworms that moved, in one launch (Swarm::update_luciferin, src/swarm.rs:66-70).
    pub fn energy_batch(&self, poses: &[f64]) -> Vec<f64> {
        let stride = self.pose_len();
        assert_eq!(poses.len() % stride, 0);
        let n = poses.len() / stride;
        let mut out = vec![0.0; n];
        let rc = unsafe { ld_scorer_energy_batch(self.handle, n, poses.as_ptr(), stride, out.as_mut_ptr()) };
        assert_eq!(rc, 0, "{}", last_error());
        out
    }
}

impl Drop for HipScore {
    fn drop(&mut self) {
        unsafe { ld_scorer_destroy(self.handle) }
    }
}

/// `GSO` (src/lib.rs:21-58) for `n_swarms` independent swarms on the device.
pub struct HipGso<'a> {
    handle: *mut c_void,
    _scorer: &'a HipScore,
}

impl<'a> HipGso<'a> {
    pub fn new(scorer: &'a HipScore, n_swarms: usize, n_glowworms: usize, positions: &[f64], seeds: Option<&[u64]>) -> Self {
        assert_eq!(positions.len(), n_swarms * n_glowworms * scorer.pose_len());
        let handle = unsafe {
            ld_gso_create(scorer.handle, n_swarms, n_glowworms, positions.as_ptr(), seeds.map_or(std::ptr::null(), |s| s.as_ptr()))
        };
        if handle.is_null() {
            panic!("{}", last_error());
        }
        HipGso { handle, _scorer: scorer }
    }

    /// GSO::run (src/lib.rs:46-58) incl. the save cadence (step 1 and every 10th).
    pub fn run(&mut self, steps: u32, output_directories: &[String]) {
        let mut done = 0u32;
        while done < steps {
            let next = if done == 0 { 1 } else { steps.min((done / 10 + 1) * 10) };
            assert_eq!(unsafe { ld_gso_run(self.handle, next - done) }, 0, "{}", last_error());
            done = next;
            if done % 10 == 0 || done == 1 {
                // one device read for all swarms, files written by threads (ld_gso_save_many); a single
                // swarm goes through ld_gso_save, the 1:1 counterpart of Swarm::save (src/swarm.rs:128-167)
                if output_directories.len() == 1 {
                    let c = CString::new(output_directories[0].as_str()).unwrap();
                    if unsafe { ld_gso_save(self.handle, 0, done, c.as_ptr()) } != 0 {
                        panic!("Error saving GSO output: {:?}", last_error());
                    }
                } else {
                    let owned: Vec<CString> = output_directories.iter().map(|d| CString::new(d.as_str()).unwrap()).collect();
                    let ptrs: Vec<*const c_char> = owned.iter().map(|c| c.as_ptr()).collect();
                    let ids: Vec<usize> = (0..owned.len()).collect();
                    if unsafe { ld_gso_save_many(self.handle, ids.len(), ids.as_ptr(), ptrs.as_ptr(), done) } != 0 {
                        panic!("Error saving GSO output: {:?}", last_error());
                    }
                }
            }
        }
    }
}

impl<'a> Drop for HipGso<'a> {
    fn drop(&mut self) {
        unsafe { ld_gso_destroy(self.handle) }
    }
}
