!  Drop-in replacement for the reference's module rrtmg_lw_init (src/rrtmg_lw_init.f90:47):
!      use rrtmg_lw_init, only: rrtmg_lw_ini
!      call rrtmg_lw_ini(cpdair)
!  forwards to rrtmg_lw_hip_init of librrtmg_lw_hip.so (include/rrtmg_lw_hip.h).  The table files are
!  looked up through the environment (RRTMG_LW_STATIC_TABLES, RRTMG_LW_KDATA, RRTMG_LW_DEVICE, RRTMG_LW_NDEV), falling
!  back to ./lw_static.bin and ./rrtmg_lw.kdata.bin - the reference's netCDF reader likewise opens the
!  literal 'rrtmg_lw.nc' in the working directory (src/rrtmg_lw_read_nc.f90:58).
      module rrtmg_lw_init

      use iso_c_binding
      use parkind, only : im => kind_im, rb => kind_rb

      implicit none

      interface
         function rrtmg_lw_hip_init(static_path, kdata_path, cpdair, device) bind(C, name='rrtmg_lw_hip_init') result(rc)
            import :: c_char, c_double, c_int
            character(kind=c_char), intent(in) :: static_path(*), kdata_path(*)
            real(c_double), value :: cpdair
            integer(c_int), value :: device
            integer(c_int) :: rc
         end function rrtmg_lw_hip_init
         function rrtmg_lw_hip_init_devices(static_path, kdata_path, cpdair, ndev, devices) bind(C, name='rrtmg_lw_hip_init_devices') result(rc)
            import :: c_char, c_double, c_int
            character(kind=c_char), intent(in) :: static_path(*), kdata_path(*)
            real(c_double), value :: cpdair
            integer(c_int), value :: ndev
            integer(c_int), intent(in) :: devices(*)
            integer(c_int) :: rc
         end function rrtmg_lw_hip_init_devices
         function rrtmg_lw_hip_host_register(ptr, bytes) bind(C, name='rrtmg_lw_hip_host_register') result(rc)
            import :: c_ptr, c_long_long, c_int
            type(c_ptr), value :: ptr
            integer(c_long_long), value :: bytes
            integer(c_int) :: rc
         end function rrtmg_lw_hip_host_register
         function rrtmg_lw_hip_host_unregister(ptr) bind(C, name='rrtmg_lw_hip_host_unregister') result(rc)
            import :: c_ptr, c_int
            type(c_ptr), value :: ptr
            integer(c_int) :: rc
         end function rrtmg_lw_hip_host_unregister
         function rrtmg_lw_hip_host_static(ptr, bytes) bind(C, name='rrtmg_lw_hip_host_static') result(rc)
            import :: c_ptr, c_long_long, c_int
            type(c_ptr), value :: ptr
            integer(c_long_long), value :: bytes
            integer(c_int) :: rc
         end function rrtmg_lw_hip_host_static
         function rrtmg_lw_hip_host_changed(ptr, keep) bind(C, name='rrtmg_lw_hip_host_changed') result(rc)
            import :: c_ptr, c_int
            type(c_ptr), value :: ptr
            integer(c_int), value :: keep
            integer(c_int) :: rc
         end function rrtmg_lw_hip_host_changed
         function rrtmg_lw_hip_gpoints() bind(C, name='rrtmg_lw_hip_gpoints') result(n)
            import :: c_int
            integer(c_int) :: n        ! ngptlw of the linked library: 140, or 256 (librrtmg_lw_hip_g256.so)
         end function rrtmg_lw_hip_gpoints
         function rrtmg_lw_hip_last_error() bind(C, name='rrtmg_lw_hip_last_error') result(p)
            import :: c_ptr
            type(c_ptr) :: p
         end function rrtmg_lw_hip_last_error
      end interface

      contains

      subroutine rrtmg_lw_ini(cpdair)
      real(kind=rb), intent(in) :: cpdair     ! Specific heat capacity of dry air at constant pressure at 273 K (J kg-1 K-1)
      character(len=1024) :: spath, kpath, dev
      integer :: ls, lk, ld, idev, ios, ndev, i
      logical :: virtual
      integer(c_int) :: rc
      integer(c_int) :: devices(16)

      call get_environment_variable('RRTMG_LW_STATIC_TABLES', spath, ls)
      if (ls == 0) then
         spath = 'lw_static.bin'; ls = 13
      endif
      call get_environment_variable('RRTMG_LW_KDATA', kpath, lk)
      if (lk == 0) then
         kpath = 'rrtmg_lw.kdata.bin'; lk = 18
      endif
      idev = 0
      call get_environment_variable('RRTMG_LW_DEVICE', dev, ld)
      if (ld > 0) then
         read(dev(1:ld), *, iostat=ios) idev
         if (ios /= 0) idev = 0
      endif
!  RRTMG_LW_NDEV = n > 1: this process drives n GPUs (ordinals RRTMG_LW_DEVICE, +1, ...); rrtmg_lw then splits its columns over
!  them, one host thread per device.  RRTMG_LW_VIRTUAL_DEVICES = 1 keeps all n on RRTMG_LW_DEVICE (separate workspaces / streams).
      ndev = 1
      call get_environment_variable('RRTMG_LW_NDEV', dev, ld)
      if (ld > 0) then
         read(dev(1:ld), *, iostat=ios) ndev
         if (ios /= 0) ndev = 1
      endif
      ndev = max(1, min(ndev, 16))
      virtual = .false.
      call get_environment_variable('RRTMG_LW_VIRTUAL_DEVICES', dev, ld)
      if (ld > 0) virtual = dev(1:1) == '1'
      if (ndev > 1) then
         do i = 1, ndev
            devices(i) = int(idev, c_int)
            if (.not. virtual) devices(i) = int(idev + i - 1, c_int)
         enddo
         rc = rrtmg_lw_hip_init_devices(spath(1:ls)//c_null_char, kpath(1:lk)//c_null_char, real(cpdair, c_double), int(ndev, c_int), devices)
      else
         rc = rrtmg_lw_hip_init(spath(1:ls)//c_null_char, kpath(1:lk)//c_null_char, real(cpdair, c_double), int(idev, c_int))
      endif
      if (rc /= 0) call rrtmg_lw_hip_abort('rrtmg_lw_ini')
      end subroutine rrtmg_lw_ini

!  Optional, for a host model whose profile / flux arrays live for the whole run: page-lock an array once (after rrtmg_lw_ini), and
!  rrtmg_lw's copies of it are direct DMA - no packing through the library's pinned staging by host threads.
!      call rrtmg_lw_pin(play, size(play))   ...   call rrtmg_lw_unpin(play)      (before the array is deallocated)
!  `a` is a whole, contiguous real(rb) array of any rank (assumed rank: the dummy is the caller's array itself, never a compiler-made
!  copy - a section or an expression would be registered at the address of a temporary that is gone on return), `n` its number of
!  elements (optional check).  A non-contiguous actual stops the run with a message instead of pinning the wrong memory.
      subroutine rrtmg_lw_pin(a, n)
      real(kind=rb), intent(in), target :: a(..)
      integer, intent(in), optional :: n
      if (size(a) < 1) return
      if (.not. is_contiguous(a)) then
         write(*,*) 'rrtmg_lw_pin: the array is not contiguous (pass the whole array, not a section)'
         error stop 1
      endif
      if (present(n)) then
         if (n /= size(a)) then
            write(*,*) 'rrtmg_lw_pin: n differs from the size of the array'
            error stop 1
         endif
      endif
      if (rrtmg_lw_hip_host_register(c_loc(a), int(size(a), c_long_long) * 8_c_long_long) /= 0) call rrtmg_lw_hip_abort('rrtmg_lw_pin')
      end subroutine rrtmg_lw_pin

      subroutine rrtmg_lw_unpin(a)
      real(kind=rb), intent(in), target :: a(..)
      if (.not. is_contiguous(a)) then
         write(*,*) 'rrtmg_lw_unpin: the array is not contiguous (pass the whole array that was pinned)'
         error stop 1
      endif
      if (rrtmg_lw_hip_host_unregister(c_loc(a)) /= 0) call rrtmg_lw_hip_abort('rrtmg_lw_unpin')
      end subroutine rrtmg_lw_unpin

!  Optional: declare an input array of rrtmg_lw STATIC - it keeps its contents from call to call (well-mixed gases, aerosol optical depths
!  that are set once, emissivities): rrtmg_lw then looks at its rows once instead of on every call.  After changing such an array:
!  call rrtmg_lw_changed(a); before deallocating it: call rrtmg_lw_changed(a, .false.).  `a` is the whole, contiguous array.
      subroutine rrtmg_lw_static(a)
      real(kind=rb), intent(in), target :: a(..)
      if (size(a) < 1) return
      if (.not. is_contiguous(a)) then
         write(*,*) 'rrtmg_lw_static: the array is not contiguous (pass the whole array, not a section)'
         error stop 1
      endif
      if (rrtmg_lw_hip_host_static(c_loc(a), int(size(a), c_long_long) * 8_c_long_long) /= 0) call rrtmg_lw_hip_abort('rrtmg_lw_static')
      end subroutine rrtmg_lw_static

      subroutine rrtmg_lw_changed(a, keep)
      real(kind=rb), intent(in), target :: a(..)
      logical, intent(in), optional :: keep
      integer(c_int) :: k
      k = 1_c_int
      if (present(keep)) then
         if (.not. keep) k = 0_c_int
      endif
      if (.not. is_contiguous(a)) then
         write(*,*) 'rrtmg_lw_changed: the array is not contiguous (pass the whole array that was declared)'
         error stop 1
      endif
      if (rrtmg_lw_hip_host_changed(c_loc(a), k) /= 0) call rrtmg_lw_hip_abort('rrtmg_lw_changed')
      end subroutine rrtmg_lw_changed

!  Turns a non-zero status of the C ABI into the reference's behaviour: print the message and stop.
      subroutine rrtmg_lw_hip_abort(where)
      character(len=*), intent(in) :: where
      type(c_ptr) :: p
      character(kind=c_char), pointer :: s(:)
      character(len=256) :: msg
      integer :: i
      msg = ' '
      p = rrtmg_lw_hip_last_error()
      if (c_associated(p)) then
         call c_f_pointer(p, s, [256])
         do i = 1, 256
            if (s(i) == c_null_char) exit
            msg(i:i) = s(i)
         enddo
      endif
      write(*,'(a,a,a)') where, ': ', trim(msg)
      error stop 1
      end subroutine rrtmg_lw_hip_abort

      end module rrtmg_lw_init
