!> The parallel layer of the dl_esm_inf API on MI355X:
!!   parallel_utils_mod  - third message-passing backend (after the reference's MPI and
!!                         serial-stub ones): rank bookkeeping from the launcher's environment,
!!                         RCCL communicator inside libdlesm_hip.so.
!!   parallel_comms_mod  - the halo-exchange message tables and exchange_generic.
!!   parallel_mod        - parallel_init, go_decompose, on_master.
!! Public names follow the reference (finite_difference/src/parallel/parallel_utils_mod.f90:62-68,
!! parallel_comms_mod.f90:153-174, parallel_mod.f90:42-47); the integer algorithms themselves
!! live in the C-ABI library so that Fortran and Python drivers share one implementation.

module parallel_utils_mod
  use iso_c_binding
  use kind_params_mod, only: go_wp
  use dlesm_hip_mod
  implicit none
  private

  integer :: rank = 1     !< 1-based, as parallel_utils_mod.f90:84
  integer :: nranks = 1
  integer :: local_rank = 0
  logical :: comm_up = .false.
  character(len=512) :: rendezvous_file = ''
  logical :: publishing = .false.   !< this rank wrote the rendezvous file and must remove it

  integer, parameter :: MSG_UNDEFINED = -99
  integer, parameter :: MSG_REQUEST_NULL = 0
  !> True once more than one rank takes part (set by parallel_init). The reference fixes it
  !! at build time (MPI=yes/no); here one library serves both cases.
  logical, protected :: DIST_MEM_ENABLED = .false.

  public parallel_init, parallel_finalise, parallel_abort, get_max_tag
  public get_rank, get_num_ranks, post_receive, post_send, global_sum
  public msg_wait, msg_wait_all, gather
  public MSG_UNDEFINED, MSG_REQUEST_NULL, DIST_MEM_ENABLED

contains

  !> First integer found among the named environment variables, else default
  integer function env_int(names, default) result(val)
    character(len=*), intent(in) :: names(:)
    integer, intent(in) :: default
    character(len=32) :: buf
    integer :: i, stat, ios
    val = default
    do i = 1, size(names)
       call get_environment_variable(trim(names(i)), buf, status=stat)
       if (stat == 0 .and. len_trim(buf) > 0) then
          read(buf, *, iostat=ios) val
          if (ios == 0) return
          val = default
       end if
    end do
  end function env_int

  logical function env_set(name)
    character(len=*), intent(in) :: name
    character(len=8) :: buf
    integer :: stat
    call get_environment_variable(name, buf, status=stat)
    env_set = (stat == 0 .and. len_trim(buf) > 0 .and. trim(buf) /= '0')
  end function env_set

  logical function env_is(name, value)
    character(len=*), intent(in) :: name, value
    character(len=8) :: buf
    integer :: stat
    call get_environment_variable(name, buf, status=stat)
    env_is = (stat == 0 .and. trim(buf) == value)
  end function env_is

  !> Ranks come from the process launcher (torchrun, mpirun, srun: whatever exported them),
  !! one process per GPU; the device is chosen from the node-local rank (the HIP counterpart
  !! of acc_init in the reference's gocean_initialise).
  subroutine parallel_init()
    integer :: ndev
    integer(c_int) :: rc
    rank = 1 + env_int([character(len=24) :: 'RANK', 'OMPI_COMM_WORLD_RANK', 'PMI_RANK', &
                        'SLURM_PROCID'], 0)
    nranks = env_int([character(len=24) :: 'WORLD_SIZE', 'OMPI_COMM_WORLD_SIZE', 'PMI_SIZE', &
                      'SLURM_NTASKS'], 1)
    local_rank = env_int([character(len=28) :: 'LOCAL_RANK', 'OMPI_COMM_WORLD_LOCAL_RANK', &
                          'SLURM_LOCALID'], rank - 1)
    if (rank < 1 .or. rank > nranks) call parallel_abort('parallel_init: inconsistent RANK/WORLD_SIZE')
    DIST_MEM_ENABLED = nranks > 1
    if (rank == 1) then
       if (nranks > 1) then
          write (*, "('Number of ranks (one GPU each):', I4)") nranks
       else
          write (*, *) 'parallel_init: single rank'
       end if
    end if
    ndev = dlesm_device_count()
    if (ndev > 0) then
       rc = dlesm_init(int(mod(local_rank, ndev), c_int))
       if (rc /= 0) call parallel_abort('parallel_init: ' // dlesm_error_text())
    end if
    ! DLESM_DRY_COMMS: keep the rank bookkeeping (decomposition, message tables) but create no
    ! communicator -- used to inspect tables on machines without GPUs.  DLESM_DRY_COMMS=2 runs
    ! the file rendezvous of the id too (with a blank id), which is how the bootstrap is tested
    ! without GPUs.
    if (nranks > 1 .and. (.not. env_set('DLESM_DRY_COMMS') .or. env_is('DLESM_DRY_COMMS', '2'))) &
         call bootstrap_rccl()
  end subroutine parallel_init

  !> Rank 0 creates the RCCL unique id and publishes it through a file (default in /dev/shm,
  !! keyed by MASTER_PORT); the others wait for it. No MPI is needed -- this is what MPI_Init
  !! does for the reference (parallel/parallel_utils_mod.f90:77-90).  The record carries a job
  !! token (world size + the launcher's run id when it exports one) and the publisher's start
  !! time; readers ignore records of other jobs, so a file left by a dead job cannot feed an
  !! old id to ncclCommInitRank.  Publishing is an in-process write + rename(2) inside
  !! libdlesm_hip.so: no child process is started once the GPU is initialised.
  subroutine bootstrap_rccl()
    character(kind=c_char) :: id(DLESM_UNIQUE_ID_BYTES)
    character(len=16) :: port
    character(len=64) :: runid
    character(len=112) :: token
    integer :: stat, timeout_s
    integer(c_int) :: rc
    call get_environment_variable('DLESM_RENDEZVOUS', rendezvous_file, status=stat)
    if (stat /= 0 .or. len_trim(rendezvous_file) == 0) then
       call get_environment_variable('MASTER_PORT', port, status=stat)
       if (stat /= 0 .or. len_trim(port) == 0) port = 'default'
       rendezvous_file = '/dev/shm/dlesm_rccl_id_' // trim(port)
    end if
    runid = env_str([character(len=24) :: 'DLESM_JOB_ID', 'TORCHELASTIC_RUN_ID', 'SLURM_JOB_ID', &
                     'PBS_JOBID', 'LSB_JOBID'], '-')
    write (token, "(I0,':',A)") nranks, trim(runid)
    timeout_s = env_int([character(len=28) :: 'DLESM_RENDEZVOUS_TIMEOUT_S'], 120)
    if (rank == 1) then
       rc = dlesm_rendezvous_remove(c_string(rendezvous_file))
       if (rc /= 0) call parallel_abort('parallel_init: ' // dlesm_error_text())
       publishing = .true.
       if (env_set('DLESM_DRY_COMMS')) then
          id = c_null_char      ! table-inspection mode: exercise the rendezvous, create no communicator
       else if (env_is('DLESM_TRANSPORT', 'mailbox')) then
          rc = dlesm_board_nonce(id)          ! a session name instead of an RCCL id
          if (rc /= 0) call parallel_abort('parallel_init: ' // dlesm_error_text())
       else
          rc = dlesm_comm_unique_id(id)
          if (rc /= 0) call parallel_abort('parallel_init: ' // dlesm_error_text())
       end if
       rc = dlesm_rendezvous_publish(c_string(rendezvous_file), id, c_string(token))
       if (rc /= 0) call parallel_abort('parallel_init: ' // dlesm_error_text())
    else
       rc = dlesm_rendezvous_fetch(c_string(rendezvous_file), id, c_string(token), &
                                   int(timeout_s * 1000, c_int))
       if (rc /= 0) call parallel_abort('parallel_init: ' // dlesm_error_text())
    end if
    if (env_set('DLESM_DRY_COMMS')) then
       ! no ncclCommInitRank to keep rank 0 here until everybody has read the record
       if (rank == 1) then
          rc = dlesm_rendezvous_wait_acks(c_string(rendezvous_file), int(nranks, c_int), &
                                          int(timeout_s * 1000, c_int))
       else
          rc = dlesm_rendezvous_ack(c_string(rendezvous_file), int(rank - 1, c_int))
       end if
       if (rc /= 0) call parallel_abort('parallel_init: ' // dlesm_error_text())
       return
    end if
    ! DLESM_TRANSPORT=mailbox: no communication library -- message plans connect their mailboxes when they are made and
    ! every exchange is stores into the neighbours' memory (dlesm_comm_init_mailbox, include/dlesm_hip.h)
    if (env_is('DLESM_TRANSPORT', 'mailbox')) then
       rc = dlesm_comm_init_mailbox(id, int(nranks, c_int), int(rank - 1, c_int))
    else
       rc = dlesm_comm_init(id, int(nranks, c_int), int(rank - 1, c_int))
    end if
    if (rc /= 0) call parallel_abort('parallel_init: ' // dlesm_error_text())
    comm_up = .true.
  end subroutine bootstrap_rccl

  !> NUL-terminated copy for the C side
  function c_string(str) result(cs)
    character(len=*), intent(in) :: str
    character(kind=c_char) :: cs(len_trim(str) + 1)
    integer :: i
    do i = 1, len_trim(str)
       cs(i) = str(i:i)
    end do
    cs(len_trim(str) + 1) = c_null_char
  end function c_string

  !> First non-empty value among the named environment variables, else default
  function env_str(names, default) result(val)
    character(len=*), intent(in) :: names(:), default
    character(len=64) :: val
    integer :: i, stat
    do i = 1, size(names)
       call get_environment_variable(trim(names(i)), val, status=stat)
       if (stat == 0 .and. len_trim(val) > 0) return
    end do
    val = default
  end function env_str

  subroutine parallel_finalise()
    integer(c_int) :: rc
    if (comm_up) then
       rc = dlesm_comm_finalize()
       comm_up = .false.
    end if
    if (publishing) then
       rc = dlesm_rendezvous_remove(c_string(rendezvous_file))
       publishing = .false.
    end if
    rc = dlesm_finalize()
  end subroutine parallel_finalise

  !> Fatal stop: message to stderr, then the process ends with a non-zero status (every rank
  !! of a job runs the same code, so they all stop; the launcher reaps stragglers).
  subroutine parallel_abort(msg)
    use iso_fortran_env, only: error_unit
    character(len=*), intent(in) :: msg
    integer(c_int) :: rc
    write(error_unit, *) msg
    flush(error_unit)
    ! do not leave this job's id file behind for the next job on the same port to trip over
    if (publishing) rc = dlesm_rendezvous_remove(c_string(rendezvous_file))
    ! mailbox mode: tell the ranks that are waiting for this one on the board (a no-op otherwise)
    rc = dlesm_board_abort(c_string(msg))
    error stop 1
  end subroutine parallel_abort

  function get_rank()
    integer :: get_rank
    get_rank = rank
  end function get_rank

  function get_num_ranks() result(num)
    integer :: num
    num = nranks
  end function get_num_ranks

  integer function get_max_tag()
    get_max_tag = 32767
  end function get_max_tag

  ! Host-buffer point-to-point calls: the device-resident exchange (exchange_generic /
  ! halo_exchange on top of dlesm_halo_exchange_f64) replaces them; they are kept in the
  ! interface for source compatibility and stop like the reference's serial stubs do.
  subroutine post_receive(nrecv, source, tag, exch_flag, rbuff, ibuff)
    real(kind=go_wp), dimension(:), optional, intent(inout) :: rbuff
    integer, dimension(:), optional, intent(inout) :: ibuff
    integer, intent(in) :: nrecv, tag, source
    integer :: exch_flag
    call parallel_abort('post_receive: host-buffer messages are not part of the RCCL backend; ' // &
                        'use exchange_generic / halo_exchange')
  end subroutine post_receive

  subroutine post_send(sendBuff, nsend, destination, tag, exch_flag)
    integer, intent(in) :: nsend, destination
    real(kind=go_wp), dimension(nsend), intent(in) :: sendBuff
    integer :: tag, exch_flag
    call parallel_abort('post_send: host-buffer messages are not part of the RCCL backend; ' // &
                        'use exchange_generic / halo_exchange')
  end subroutine post_send

  subroutine msg_wait(nmsg, flags, irecv)
    integer, intent(in) :: nmsg
    integer, dimension(:), intent(inout) :: flags
    integer, intent(out) :: irecv
    irecv = MSG_UNDEFINED   ! nothing outstanding: exchanges are stream-ordered
  end subroutine msg_wait

  subroutine msg_wait_all(nmsg, flags)
    integer, intent(in) :: nmsg
    integer, dimension(:), intent(inout) :: flags
  end subroutine msg_wait_all

  !> Sum of one fp64 scalar over all ranks (ncclAllReduce); no-op on one rank.
  subroutine global_sum(var)
    real(go_wp), intent(inout) :: var
    integer(c_int) :: rc
    real(c_double) :: v
    if (nranks == 1) return
    v = var
    rc = dlesm_global_sum_f64(v)
    if (rc /= 0) call parallel_abort('global_sum: ' // dlesm_error_text())
    var = v
  end subroutine global_sum

  !> Gather size(send_buffer) values per rank onto rank 1 (root 0), staged through the device
  !! because RCCL moves device memory.
  subroutine gather(send_buffer, recv_buffer)
    real(go_wp), dimension(:), target :: send_buffer, recv_buffer
    type(c_ptr) :: dsend, drecv
    integer(c_size_t) :: nb
    integer(c_int) :: rc
    integer :: n
    if (nranks == 1) then
       recv_buffer = send_buffer
       return
    end if
    n = size(send_buffer)
    nb = int(n, c_size_t) * 8_c_size_t
    drecv = c_null_ptr
    if (hipMalloc(dsend, nb) /= 0) call parallel_abort('gather: hipMalloc failed')
    if (hipMemcpy(dsend, c_loc(send_buffer), nb, 1_c_int) /= 0) call parallel_abort('gather: H2D failed')
    if (rank == 1) then
       if (hipMalloc(drecv, nb * int(nranks, c_size_t)) /= 0) call parallel_abort('gather: hipMalloc failed')
    end if
    rc = dlesm_gather_f64(dsend, drecv, int(n, c_int))
    if (rc /= 0) call parallel_abort('gather: ' // dlesm_error_text())
    if (rank == 1) then
       if (hipMemcpy(c_loc(recv_buffer), drecv, nb * int(nranks, c_size_t), 2_c_int) /= 0) &
            call parallel_abort('gather: D2H failed')
       rc = hipFree(drecv)
    end if
    rc = hipFree(dsend)
  end subroutine gather

end module parallel_utils_mod


module parallel_comms_mod
  use iso_c_binding
  use kind_params_mod, only: go_wp
  use parallel_utils_mod, only: get_num_ranks, get_rank, parallel_abort, global_sum
  use decomposition_mod, only: subdomain_type, decomposition_type
  use dlesm_hip_mod
  implicit none
  private

  integer, parameter :: MAX_HALO_DEPTH = 1
  !> deepest exchange the device layer builds tables for (dlesm_map_comms_depth)
  integer, parameter :: MAX_DEVICE_HALO_DEPTH = 8
  integer, parameter :: MaxComm = DLESM_MAXCOMM

  ! One rank's message lists (reference parallel_comms_mod.f90:52-83); a module-level
  ! singleton like there: one decomposition per process.
  integer, save, dimension(MaxComm) :: dirsend, destination, dirrecv, source
  integer, save, dimension(MaxComm) :: isrcsend, jsrcsend, isrcrecv, jsrcrecv, &
       idessend, jdessend, nxsend, nysend, nzsend, idesrecv, jdesrecv, nxrecv, nyrecv, nzrecv
  integer, save :: nsend = 0, nrecv = 0
  integer, save, dimension(MaxComm, MAX_HALO_DEPTH) :: nsendp, nsendp2d, nrecvp, nrecvp2d
  integer, save :: ielb, ieub, iesub, jelb, jeub, jesub

  ! direction codes (parallel_comms_mod.f90:101-110)
  integer, parameter :: NONE = 0, Iplus = 1, Iminus = 2, Jplus = 3, Jminus = 4, &
       IplusJplus = 5, IminusJminus = 6, IplusJminus = 7, IminusJplus = 8, MaxCommDir = 8
  integer, dimension(MaxCommDir) :: opp_dirn = (/ Iminus, Iplus, Jminus, Jplus, &
       IminusJminus, IplusJplus, IminusJplus, IplusJminus /)

  ! the C-side copy of the tables and the device message plans built from them
  type(c_comm_tables), save :: ctables
  logical, save :: have_tables = .false.
  type(c_ptr), save :: plan = c_null_ptr, scratch = c_null_ptr
  integer, save :: plan_ld = 0, plan_ny = 0

  public :: map_comms, iprocmap, exchmod_alloc, exchange_generic, global_sum
  public :: MaxComm, nsend, nrecv, nxsend, nysend, destination, dirrecv, dirsend, isrcsend, &
            jsrcsend, idesrecv, jdesrecv, nxrecv, nyrecv, source, idessend, jdessend
  public :: nsendp, nsendp2d, nrecvp, nrecvp2d
  public :: ielb, ieub, jeub, jelb
  public :: NONE, Iplus, Iminus, Jplus, Jminus, IplusJplus, IminusJminus, IplusJminus, &
            IminusJplus, MaxCommDir
  public :: opp_dirn
  ! additions of this implementation
  public :: halo_plan_for, exchange_device, to_c_decomp

contains

  subroutine to_c_decomp(decomp, cd, csubs)
    type(decomposition_type), intent(in) :: decomp
    type(c_decomp), intent(out) :: cd
    type(c_subdomain), allocatable, intent(out) :: csubs(:)
    integer :: i
    cd%global_nx = decomp%global_nx;  cd%global_ny = decomp%global_ny
    cd%nx = decomp%nx;  cd%ny = decomp%ny
    cd%ndomains = decomp%ndomains
    cd%max_width = decomp%max_width;  cd%max_height = decomp%max_height
    allocate(csubs(decomp%ndomains))
    do i = 1, decomp%ndomains
       associate (g => decomp%subdomains(i)%global, t => decomp%subdomains(i)%internal)
         csubs(i)%global = c_region(g%nx, g%ny, g%xstart, g%xstop, g%ystart, g%ystop)
         csubs(i)%internal = c_region(t%nx, t%ny, t%xstart, t%xstop, t%ystart, t%ystop)
       end associate
    end do
  end subroutine to_c_decomp

  !> Build this rank's send/receive lists (reference parallel_comms_mod.f90:178-1172).
  subroutine map_comms(decomp, tmask, pbc, halo_depths, ierr)
    use parallel_utils_mod, only: DIST_MEM_ENABLED
    type(decomposition_type), target, intent(in) :: decomp
    integer, intent(in), allocatable :: tmask(:,:)
    logical, intent(in) :: pbc
    integer, intent(in) :: halo_depths(2)
    integer, intent(out) :: ierr
    type(c_decomp) :: cd
    type(c_subdomain), allocatable :: csubs(:)
    integer :: irank
    integer(c_int) :: rc

    ierr = 0
    if (.not. DIST_MEM_ENABLED) return
    ! The reference aborts beyond MAX_HALO_DEPTH = 1.  Extension of this layer: equal depths
    ! 2..MAX_DEVICE_HALO_DEPTH select the deep tables the fused multi-step kernels need.
    if (halo_depths(1) /= halo_depths(2) .or. halo_depths(1) > MAX_DEVICE_HALO_DEPTH) then
       call parallel_abort('map_comms: specified halo depth exceeds MAX_HALO_DEPTH limit in ' // &
                           'parallel_comms_mod')
    end if
    if (pbc) call parallel_abort('map_comms: periodic-boundary conditions are not yet supported')

    irank = get_rank()
    call to_c_decomp(decomp, cd, csubs)
    if (halo_depths(1) > MAX_HALO_DEPTH) then
       rc = dlesm_map_comms_depth(cd, csubs, int(get_num_ranks(), c_int), int(irank, c_int), &
                                  int(halo_depths(1), c_int), ctables)
    else
       rc = dlesm_map_comms(cd, csubs, int(get_num_ranks(), c_int), int(irank, c_int), ctables)
    end if
    if (rc == DLESM_ECOMMS) then
       ierr = -12
       return
    else if (rc /= 0) then
       call parallel_abort('map_comms: ' // dlesm_error_text())
    end if
    have_tables = .true.
    call drop_plans()

    nsend = ctables%nsend;  nrecv = ctables%nrecv
    dirsend = ctables%dirsend;  destination = ctables%destination
    isrcsend = ctables%isrcsend;  jsrcsend = ctables%jsrcsend
    idessend = ctables%idessend;  jdessend = ctables%jdessend
    nxsend = ctables%nxsend;  nysend = ctables%nysend
    dirrecv = ctables%dirrecv;  source = ctables%source
    isrcrecv = ctables%isrcrecv;  jsrcrecv = ctables%jsrcrecv
    idesrecv = ctables%idesrecv;  jdesrecv = ctables%jdesrecv
    nxrecv = ctables%nxrecv;  nyrecv = ctables%nyrecv
    nzsend = -999;  nzrecv = -999
    nzsend(1:nsend) = 1;  nzrecv(1:nrecv) = 1
    nsendp2d(:, 1) = nxsend * nysend;  nsendp(:, 1) = nsendp2d(:, 1)
    nrecvp2d(:, 1) = nxrecv * nyrecv;  nrecvp(:, 1) = nrecvp2d(:, 1)

    associate (me => decomp%subdomains(irank))
      jelb = me%global%ystart;  jeub = me%global%ystop
      ielb = me%global%xstart;  ieub = me%global%xstop
      iesub = me%internal%nx;   jesub = me%internal%ny
    end associate
  end subroutine map_comms

  !> 1-based owner of global point (ia, ja), 0 if none (reference :1365-1398)
  function iprocmap(decomp, ia, ja)
    integer :: iprocmap
    type(decomposition_type), intent(in) :: decomp
    integer, intent(in) :: ia, ja
    type(c_decomp) :: cd
    type(c_subdomain), allocatable :: csubs(:)
    call to_c_decomp(decomp, cd, csubs)
    iprocmap = dlesm_iprocmap(cd, csubs, int(get_num_ranks(), c_int), int(ia, c_int), int(ja, c_int))
  end function iprocmap

  !> Kept for source compatibility: the flag/tag pools of the MPI exchange are gone
  !! (stream ordering replaces them).
  integer function exchmod_alloc()
    exchmod_alloc = 0
  end function exchmod_alloc

  subroutine drop_plans()
    integer(c_int) :: rc
    if (c_associated(plan)) rc = dlesm_halo_plan_destroy(plan)
    if (c_associated(scratch)) rc = dlesm_field_destroy(scratch)
    plan = c_null_ptr;  scratch = c_null_ptr
    plan_ld = 0;  plan_ny = 0
  end subroutine drop_plans

  !> Device message plan for fields of extent (ld, ny): built once, shared by every field of
  !! the grid (all fields are allocated with the grid's extents).
  function halo_plan_for(ld, ny) result(p)
    integer, intent(in) :: ld, ny
    type(c_ptr) :: p
    integer(c_int) :: rc
    if (.not. have_tables) call parallel_abort('halo exchange before map_comms (grid_init)')
    if (c_associated(plan) .and. (ld /= plan_ld .or. ny /= plan_ny)) call drop_plans()
    if (.not. c_associated(plan)) then
       rc = dlesm_halo_plan_create(ctables, int(ld, c_int), int(ny, c_int), plan)
       if (rc /= 0) call parallel_abort('halo plan: ' // dlesm_error_text())
       plan_ld = ld;  plan_ny = ny
    end if
    p = plan
  end function halo_plan_for

  integer(c_int) function dir_mask(comm1, comm2, comm3, comm4) result(mask)
    integer, intent(in) :: comm1, comm2, comm3, comm4
    integer :: c(4), i
    c = (/ comm1, comm2, comm3, comm4 /)
    mask = 0
    do i = 1, 4
       if (c(i) >= 1 .and. c(i) <= 4) mask = ior(mask, ishft(1_c_int, c(i) - 1))
    end do
  end function dir_mask

  !> enabled(dir) of the reference (:1557-1571): edges by their bit, diagonals when both edges are
  logical function dir_enabled(mask, dir)
    integer(c_int), intent(in) :: mask
    integer, intent(in) :: dir
    select case (dir)
    case (Iplus, Iminus, Jplus, Jminus)
       dir_enabled = btest(mask, dir - 1)
    case (IplusJplus)
       dir_enabled = btest(mask, Iplus - 1) .and. btest(mask, Jplus - 1)
    case (IminusJminus)
       dir_enabled = btest(mask, Iminus - 1) .and. btest(mask, Jminus - 1)
    case (IplusJminus)
       dir_enabled = btest(mask, Iplus - 1) .and. btest(mask, Jminus - 1)
    case (IminusJplus)
       dir_enabled = btest(mask, Iminus - 1) .and. btest(mask, Jplus - 1)
    case default
       dir_enabled = .false.
    end select
  end function dir_enabled

  !> Exchange the halos of a field that already lives on the device (raw device pointer).
  subroutine exchange_device(dev_data, ld, ny, comm1, comm2, comm3, comm4)
    type(c_ptr), intent(in) :: dev_data
    integer, intent(in) :: ld, ny, comm1, comm2, comm3, comm4
    integer(c_int) :: rc
    rc = dlesm_halo_exchange_f64(halo_plan_for(ld, ny), dev_data, &
                                 dir_mask(comm1, comm2, comm3, comm4), c_null_ptr)
    if (rc /= 0) call parallel_abort('halo_exchange: ' // dlesm_error_text())
    if (hipDeviceSynchronize() /= 0) call parallel_abort('halo_exchange: device synchronisation failed')
  end subroutine exchange_device

  !> Halo exchange of a HOST array (reference :1501-1855): the strips to send go up to a
  !! device scratch field, the exchange runs device-to-device over RCCL, the received strips
  !! come back -- the mirror image of what the reference does around its host MPI exchange
  !! for device-resident fields (field_mod.f90:1241-1254).
  subroutine exchange_generic(b2, ib2, b3, ib3, handle, comm1, comm2, comm3, comm4)
    use parallel_utils_mod, only: DIST_MEM_ENABLED
    integer, intent(out) :: handle
    real(go_wp), optional, intent(inout), dimension(:,:), contiguous, target :: b2
    integer, optional, intent(inout), dimension(:,:) :: ib2
    real(go_wp), optional, intent(inout), dimension(:,:,:) :: b3
    integer, optional, intent(inout), dimension(:,:,:) :: ib3
    integer, intent(in) :: comm1, comm2, comm3, comm4
    integer :: k, ld, ny
    integer(c_int) :: rc, mask
    type(c_ptr) :: p

    handle = 0
    if (.not. DIST_MEM_ENABLED) return
    if (present(ib2)) call parallel_abort('exchange_generic: halo-swaps for 2D integer fields ' // &
                                          'not implemented.')
    if (present(b3) .or. present(ib3)) call parallel_abort('exchange_generic: halo-swaps for 3D ' // &
                                                           'fields not implemented.')
    if (.not. present(b2)) return
    ld = size(b2, 1);  ny = size(b2, 2)
    p = halo_plan_for(ld, ny)
    if (.not. c_associated(scratch)) then
       rc = dlesm_field_create(int(ld, c_int), int(ny, c_int), scratch)
       if (rc /= 0) call parallel_abort('exchange_generic: ' // dlesm_error_text())
    end if
    mask = dir_mask(comm1, comm2, comm3, comm4)
    if (mask == 0) return   ! no direction enabled: the reference posts nothing either (:1557-1571)
    ! only the messages of enabled directions travel: host halos of a disabled direction must
    ! not be overwritten with whatever the persistent scratch field holds
    do k = 1, nsend
       if (.not. dir_enabled(mask, dirsend(k))) cycle
       call dlesm_write_to_device(c_loc(b2), scratch, int(isrcsend(k), c_int), int(jsrcsend(k), c_int), &
                                  int(nxsend(k), c_int), int(nysend(k), c_int), .false._c_bool)
    end do
    rc = dlesm_transfer_sync()
    if (rc /= 0) call parallel_abort('exchange_generic: ' // dlesm_error_text())
    rc = dlesm_halo_exchange_f64(p, dlesm_field_data(scratch), mask, c_null_ptr)
    if (rc /= 0) call parallel_abort('exchange_generic: ' // dlesm_error_text())
    if (hipDeviceSynchronize() /= 0) call parallel_abort('exchange_generic: device synchronisation failed')
    do k = 1, nrecv
       if (.not. dir_enabled(mask, dirrecv(k))) cycle
       call dlesm_read_from_device(scratch, c_loc(b2), int(idesrecv(k), c_int), int(jdesrecv(k), c_int), &
                                   int(nxrecv(k), c_int), int(nyrecv(k), c_int), .false._c_bool)
    end do
    rc = dlesm_transfer_sync()
    if (rc /= 0) call parallel_abort('exchange_generic: ' // dlesm_error_text())
  end subroutine exchange_generic

end module parallel_comms_mod


module parallel_mod
  use iso_c_binding
  use parallel_utils_mod, only: parallel_finalise, parallel_abort, get_rank, get_num_ranks
  use parallel_comms_mod, only: map_comms, exchmod_alloc
  use decomposition_mod, only: decomposition_type
  use dlesm_hip_mod
  implicit none
  private
  public parallel_init, parallel_finalise, parallel_abort, go_decompose
  public on_master
  public map_comms, get_rank, get_num_ranks
  public decomposition_type

contains

  subroutine parallel_init()
    use parallel_utils_mod, only: init => parallel_init
    call init()
    if (exchmod_alloc() /= 0) call parallel_abort('Failed to allocate message buffers')
  end subroutine parallel_init

  !> Cut a domainx x domainy domain into a grid of tiles (reference parallel_mod.f90:70-332).
  function go_decompose(domainx, domainy, ndomains, ndomainx, ndomainy, halo_width) result(decomp)
    type(decomposition_type), target :: decomp
    integer, intent(in) :: domainx, domainy
    integer, intent(in), optional :: ndomains, ndomainx, ndomainy, halo_width
    integer :: ndom, tx, ty, hwidth, nranks, i
    type(c_decomp) :: cd
    type(c_subdomain), allocatable :: cs(:)
    integer(c_int) :: rc

    tx = 0;  ty = 0
    if (present(ndomains)) then
       ndom = ndomains
    else if (present(ndomainx) .and. present(ndomainy)) then
       ndom = ndomainx * ndomainy
       tx = ndomainx;  ty = ndomainy
    else if (.not. present(ndomainx) .and. .not. present(ndomainy)) then
       ndom = get_num_ranks()
    else
       call parallel_abort('go_decompose: invalid arguments supplied')
    end if
    nranks = get_num_ranks()
    if (nranks < 1) call parallel_abort('go_decompose: nranks must be >= 1 ')
    hwidth = 1
    if (present(halo_width)) then
       if (halo_width < 1 .and. nranks > 1) call parallel_abort('go_decompose: halo width must ' // &
            'be > 0 if running on more than one process')
       hwidth = halo_width
    end if

    allocate(cs(ndom))
    rc = dlesm_decompose(int(domainx, c_int), int(domainy, c_int), int(ndom, c_int), int(tx, c_int), &
                         int(ty, c_int), int(hwidth, c_int), cd, cs)
    if (rc /= 0) call parallel_abort('go_decompose: ' // dlesm_error_text())

    decomp%global_nx = cd%global_nx;  decomp%global_ny = cd%global_ny
    decomp%nx = cd%nx;  decomp%ny = cd%ny
    decomp%ndomains = cd%ndomains
    decomp%max_width = cd%max_width;  decomp%max_height = cd%max_height
    allocate(decomp%subdomains(ndom))
    do i = 1, ndom
       associate (g => decomp%subdomains(i)%global, t => decomp%subdomains(i)%internal)
         g%nx = cs(i)%global%nx;  g%ny = cs(i)%global%ny
         g%xstart = cs(i)%global%xstart;  g%xstop = cs(i)%global%xstop
         g%ystart = cs(i)%global%ystart;  g%ystop = cs(i)%global%ystop
         t%nx = cs(i)%internal%nx;  t%ny = cs(i)%internal%ny
         t%xstart = cs(i)%internal%xstart;  t%xstop = cs(i)%internal%xstop
         t%ystart = cs(i)%internal%ystart;  t%ystop = cs(i)%internal%ystop
       end associate
    end do
    ! rank n owns subdomain n; with fewer ranks than subdomains (OpenMP tiling of one
    ! field) they are dealt out in order, unused slots are -1
    allocate(decomp%proc_subdomains((ndom + nranks - 1) / nranks, nranks))
    decomp%proc_subdomains = -1
    do i = 1, ndom
       decomp%proc_subdomains(mod(i - 1, size(decomp%proc_subdomains, 1)) + 1, &
                              (i - 1) / size(decomp%proc_subdomains, 1) + 1) = i
    end do
    if (on_master()) then
       write (*, "('go_decompose: using grid of ',I3,'x',I3)") decomp%nx, decomp%ny
    end if
  end function go_decompose

  function on_master()
    logical :: on_master
    on_master = get_rank() == 1
  end function on_master

end module parallel_mod
