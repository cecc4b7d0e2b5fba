!> Basic GOcean utilities: start-up / shut-down, fatal stop, master-only logging.
!! Interface of the reference's gocean_mod (finite_difference/src/gocean_mod.F90:9-15).
!! gocean_initialise is where the reference binds an OpenACC device
!! (acc_init(acc_device_nvidia), gocean_mod.F90:31-33); here parallel_init selects the
!! HIP device of this rank.
module gocean_mod
  use kind_params_mod
  implicit none
  private

  interface model_write_log
     module procedure log_text, log_int_real, log_int, log_real
  end interface

  public gocean_initialise, gocean_finalise, gocean_stop
  public model_write_log

contains

  subroutine gocean_initialise()
    use parallel_mod, only: parallel_init
    call parallel_init()
  end subroutine gocean_initialise

  subroutine gocean_finalise()
    use parallel_mod, only: parallel_finalise
    call parallel_finalise()
  end subroutine gocean_finalise

  !> Stop the model run (fatal): every error path of the library ends here.
  subroutine gocean_stop(msg)
    use parallel_mod, only: parallel_abort
    character(len=*), intent(in) :: msg
    call parallel_abort(msg)
  end subroutine gocean_stop

  logical function speaks(all_ranks)
    use parallel_mod, only: on_master
    logical, optional, intent(in) :: all_ranks
    speaks = on_master()
    if (present(all_ranks)) speaks = speaks .or. all_ranks
  end function speaks

  subroutine log_int_real(fmtstr, istep, fvar, all_ranks)
    use iso_fortran_env, only: output_unit
    character(len=*), intent(in) :: fmtstr
    integer, intent(in) :: istep
    real(go_wp), intent(in) :: fvar
    logical, optional :: all_ranks
    if (speaks(all_ranks)) write(output_unit, fmt=fmtstr) istep, fvar
  end subroutine log_int_real

  subroutine log_int(fmtstr, istep, all_ranks)
    use iso_fortran_env, only: output_unit
    character(len=*), intent(in) :: fmtstr
    integer, intent(in) :: istep
    logical, optional :: all_ranks
    if (speaks(all_ranks)) write(output_unit, fmt=fmtstr) istep
  end subroutine log_int

  subroutine log_real(fmtstr, fvar, all_ranks)
    use iso_fortran_env, only: output_unit
    character(len=*), intent(in) :: fmtstr
    real(go_wp), intent(in) :: fvar
    logical, optional :: all_ranks
    if (speaks(all_ranks)) write(output_unit, fmt=fmtstr) fvar
  end subroutine log_real

  subroutine log_text(fmtstr, msg, all_ranks)
    use iso_fortran_env, only: output_unit
    character(len=*), intent(in) :: fmtstr
    character(len=*), intent(in) :: msg
    logical, optional :: all_ranks
    if (speaks(all_ranks)) write(output_unit, fmt=fmtstr) msg
  end subroutine log_text

end module gocean_mod
